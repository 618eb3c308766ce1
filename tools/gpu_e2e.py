"""Host-to-host (PCIe-inclusive) rate of td_encode_batch, for DESIGN.md (never the bench `value`)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from tokendagger_amd import capi, vocab_io
import bench
name, pat, ranks, special = vocab_io.load_tdv(vocab_io.default_vocab_path())
tok = capi.HipTokenizer(pat, ranks, special, device=0)
for kind, mb in [("english", 256), ("code", 64), ("mixed", 64)]:
    n = mb << 20
    x, offs = bench.build_corpus(kind, n, 1000)
    tok.encode_batch(x, offs, capacity=n)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); toks, toffs = tok.encode_batch(x, offs, capacity=n); best = min(best, time.perf_counter() - t0)
    print(f"host->host td_encode_batch {kind} {mb} MiB: {best*1e3:.1f} ms = {n/best/1e9:.2f} GB/s ({len(toks)} tokens, pageable host memory)", flush=True)
