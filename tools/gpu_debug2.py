"""Debug aid: replay the fuzz batches of test_fuzz_batches_vs_oracle and print the first differing documents."""
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
    text, offs = H.pack_docs(docs)
    toks, toffs = tok.encode_batch(text, offs, mode=it % 2)
    et, eo = O.encode_batch(text, offs)
    bad = 0
    for d in range(len(docs)):
        a = toks[toffs[d]:toffs[d + 1]]; b = et[eo[d]:eo[d + 1]]
        if not np.array_equal(a, b):
            bad += 1
            if bad <= 4:
                n = min(len(a), len(b)); k = int(np.argmax(a[:n] != b[:n])) if n and (a[:n] != b[:n]).any() else n
                print(f"it {it} doc {d} off {offs[d]} (mod 8192: {offs[d] % 8192}) len {len(docs[d])} got {len(a)} exp {len(b)} first diff {k}", a[max(0, k - 2):k + 4], b[max(0, k - 2):k + 4], repr(docs[d][:40]), repr(docs[d][-20:]))
                # single-document encode of the same doc
                s = tok.encode(docs[d], mode=it % 2)
                print("     alone:", "ok" if np.array_equal(s, b) else f"ALSO BAD ({len(s)} ids)")
    print("it", it, "docs", len(docs), "bad", bad, "long pieces", tok.info(7))
