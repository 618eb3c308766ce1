"""Host-to-host td_encode_batch on 1024 MiB of English: the pipeline's chunk size and copy threads (TD_OPT_PIPE_CHUNK_BYTES / _THREADS).
GPU box:  python tools/gpu_e2e_sweep.py"""
import ctypes, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from tokendagger_amd import capi, vocab_io
import bench
name, pat, ranks, special = vocab_io.load_tdv(vocab_io.default_vocab_path())
tok = capi.HipTokenizer(pat, ranks, special, device=0)
lib = capi.load_library()
n = 1024 << 20
x, offs = bench.build_corpus("english", n, 1000)
nd = len(offs) - 1
cap = n // 2 + 1024
toks = np.zeros(cap, dtype=np.int32); toff = np.zeros(nd + 1, dtype=np.int64); ntok = ctypes.c_int64(0)
def call():
    rc = lib.td_encode_batch(tok._h, x.ctypes.data, offs.ctypes.data, nd, 0, toks.ctypes.data, cap, toff.ctypes.data, ctypes.byref(ntok))
    assert rc == 0, rc
for threads in (16, 32):
    for mb in (16, 32, 64, 128):
        tok.set_option(capi.TD_OPT_PIPE_THREADS, threads); tok.set_option(capi.TD_OPT_PIPE_CHUNK_BYTES, mb << 20)
        call()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); call(); ts.append(time.perf_counter() - t0)
        s = statistics.median(ts)
        print(f"threads {threads:3d} chunk {mb:4d} MiB: {s * 1e3:7.2f} ms = {n / s / 1e9:6.2f} GB/s  (runs {[round(v * 1e3, 1) for v in ts]})", flush=True)
