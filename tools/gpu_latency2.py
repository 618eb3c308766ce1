import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from tokendagger_amd import capi, vocab_io
name, pat, ranks, special = vocab_io.load_tdv(vocab_io.default_vocab_path())
tok = capi.HipTokenizer(pat, ranks, special, device=0)
lib = capi.load_library()
for s in (b"Hello, world!", b"The quick brown fox jumps over the lazy dog. " * 20, b"x" * 10 + b" y" * 1500):
    x = np.frombuffer(s, dtype=np.uint8).copy(); offs = np.asarray([0, len(x)], dtype=np.int64)
    toks = np.zeros(len(x) + 16, dtype=np.int32); toff = np.zeros(2, dtype=np.int64); nt = ctypes.c_int64(0)
    f = lib.td_encode_batch; args = (tok._h, x.ctypes.data, offs.ctypes.data, 1, 0, toks.ctypes.data, len(toks), toff.ctypes.data, ctypes.byref(nt))
    for _ in range(50): f(*args)
    t0 = time.perf_counter()
    for _ in range(2000): f(*args)
    print(f"C ABI td_encode_batch {len(s)} bytes: {(time.perf_counter() - t0) / 2000 * 1e6:.1f} us per call ({nt.value} ids)", flush=True)
