import sys, random, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import helpers as H
from tokendagger_amd import capi
pat, mr, sp = H.llama4()
tok = capi.HipTokenizer(pat, mr, sp, device=0)
O = H.port_tokenizer()
rng = random.Random(2025)
for it in range(12):
    docs = []
    for _ in range(rng.randint(1, 400)):
        r = rng.random()
        if r < 0.05: docs.append(b"")
        elif r < 0.10: docs.append((rng.choice(["a", " ", "=", "1", "\n", "A", "xY", "中"]) * rng.randint(50, 9000)).encode())
        elif r < 0.5: docs.append(H.random_unicode_string(rng, 200).encode("utf-8"))
        else: docs.append("".join(H.fuzz_string(rng) for _ in range(rng.randint(1, 40))).encode("utf-8"))
    if it != 9: continue
    text, offs = H.pack_docs(docs)
    def check(lo, hi, label):
        sub = docs[lo:hi]
        base = offs[lo]
        padlen = base % 8192 + 8192 * 2
        t, o = H.pack_docs([b"x " * (padlen // 2) + b"y" * (padlen % 2)] + sub)
        toks, toffs = tok.encode_batch(t, o, mode=1)
        et, eo = O.encode_batch(t, o)
        bad = [d for d in range(len(sub) + 1) if not np.array_equal(toks[toffs[d]:toffs[d+1]], et[eo[d]:eo[d+1]])]
        print(label, lo, hi, "bad docs (1-based in sub):", bad)
        return bad
    for w in (40, 20, 10, 5, 3, 2, 1):
        check(max(0, 151 - w), 153, f"window-{w}")
    for d in range(146, 153):
        print(d, offs[d], offs[d] % 8192, len(docs[d]), repr(docs[d][:30]), repr(docs[d][-12:]))
