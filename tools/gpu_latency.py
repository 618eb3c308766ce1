"""Small-call latency of the host API (one short string per call), for DESIGN.md."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import helpers as H
import tokendagger as tiktoken
pat, mr, sp = H.llama4()
enc = tiktoken.Encoding("llama4", pat_str=pat, mergeable_ranks=mr, special_tokens=sp)
for text in ["Hello, world!", "The quick brown fox jumps over the lazy dog. " * 20,
             "The quick brown fox jumps over the lazy dog. " * 450]:  # (no giant single pieces here: a 20 KB run of one
    # letter is ONE piece, and pieces above 1 KiB merge in O(len^2 / 64) rounds in an HBM pool, as in the reference)
    for _ in range(20): enc.encode(text)
    t0 = time.perf_counter(); n = 300
    for _ in range(n): ids = enc.encode(text)
    dt = (time.perf_counter() - t0) / n
    for _ in range(20): enc.decode(ids)
    t0 = time.perf_counter()
    for _ in range(n): enc.decode(ids)
    dd = (time.perf_counter() - t0) / n
    print(f"{len(text):6d} chars: encode {dt*1e6:7.1f} us/call ({len(ids)} ids), decode {dd*1e6:7.1f} us/call", flush=True)
