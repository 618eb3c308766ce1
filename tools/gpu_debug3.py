import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import helpers as H
from tokendagger_amd import capi
pat, mr, sp = H.llama4()
tok = capi.HipTokenizer(pat, mr, sp, device=0)
O = H.port_tokenizer()
pad = (b"hello world " * 20000)[:124029]
for body in (b"1" * 7529, "中".encode() * 5218):
    docs = [pad, body, b"tail doc here"]
    text, offs = H.pack_docs(docs)
    toks, toffs = tok.encode_batch(text, offs)
    et, eo = O.encode_batch(text, offs)
    for d in range(3):
        a = toks[toffs[d]:toffs[d+1]]; b = et[eo[d]:eo[d+1]]
        ok = np.array_equal(a, b)
        print("doc", d, "ok" if ok else "BAD", len(a), len(b))
        if not ok:
            # byte position of each token via decode lengths
            la = np.cumsum([len(tok.decode_bytes([int(t)])) for t in a]); lb = np.cumsum([len(tok.decode_bytes([int(t)])) for t in b])
            n = min(len(a), len(b)); k = int(np.argmax(a[:n] != b[:n])) if (a[:n] != b[:n]).any() else n
            print("  first diff token", k, "byte in doc", (la[k-1] if k else 0), "global", offs[d] + (la[k-1] if k else 0), "mod 8192", (offs[d] + (la[k-1] if k else 0)) % 8192, a[k-2:k+5], b[k-2:k+5])
