"""Host-side rates of the drop-in surface (PCIe-inclusive; never the bench `value`):
  * C ABI td_encode_batch, host buffers in / host buffers out (output buffer allocated and touched beforehand);
  * Python Tokenizer.encode_batch(list[str]) -> list[list[int]] on the reference benchmark's chunking (T x 10 slices,
    tests/throughput_test.py:399-416) and encode_batch_to_numpy on the same bytes;
  * one-string calls: enc.encode(s) latency over the reference's performance_benchmark text shapes."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from tokendagger_amd import capi, vocab_io
import bench, td_corpus
out = {}
name, pat, ranks, special = vocab_io.load_tdv(vocab_io.default_vocab_path())
tok = capi.HipTokenizer(pat, ranks, special, device=0)
lib = capi.load_library()
for kind, mb in [("english", 256), ("english", 1024), ("code", 256), ("mixed", 256)]:
    n = mb << 20
    x, offs = bench.build_corpus(kind, n, 1000)
    nd = len(offs) - 1
    cap = n // 2 + 1024 if kind == "english" else n
    toks = np.zeros(cap, dtype=np.int32)          # allocated AND touched: first-touch page faults are the caller's, not the library's
    toff = np.zeros(nd + 1, dtype=np.int64)
    ntok = ctypes.c_int64(0)
    def call():
        rc = lib.td_encode_batch(tok._h, x.ctypes.data, offs.ctypes.data, nd, 0, toks.ctypes.data, cap, toff.ctypes.data, ctypes.byref(ntok))
        assert rc == 0, rc
    call()
    ts = []
    for _ in range(4):
        t0 = time.perf_counter(); call(); ts.append(time.perf_counter() - t0)
    best = min(ts)
    out[f"c_abi_host_to_host_{kind}_{mb}"] = {"ms": round(best * 1e3, 2), "GB/s": round(n / best / 1e9, 2), "tokens": ntok.value}
    print(f"C ABI host->host {kind} {mb} MiB: {best*1e3:.1f} ms = {n/best/1e9:.2f} GB/s ({ntok.value} tokens)", flush=True)
del toks
# ---- Python surface --------------------------------------------------------------------------------------------
import tokendagger as tiktoken
mr = dict(ranks)
for k, v in special.items():
    mr[k.encode("utf-8")] = v
enc = tiktoken.Encoding(name="llama4", pat_str=pat, mergeable_ranks=mr, special_tokens=special)
n = 256 << 20
x, _ = bench.build_corpus("english", n, 1000)
text = x.tobytes().decode("ascii")
for T in (1, 32):
    co = td_corpus.chunk_offsets(n, T * 10)
    chunks = [text[co[i]:co[i + 1]] for i in range(T * 10)]
    enc.encode_batch(chunks[:2], num_threads=T)
    t0 = time.perf_counter(); res = enc.encode_batch(chunks, num_threads=T); dt = time.perf_counter() - t0
    ntok = sum(len(r) for r in res)
    out[f"python_encode_batch_list_T{T}"] = {"s": round(dt, 3), "GB/s": round(n / dt / 1e9, 3), "MiB/s": round(256 / dt, 1), "tokens": ntok}
    print(f"Python encode_batch(list[str]) -> list[list[int]], {T*10} slices of 256 MiB: {dt:.3f} s = {n/dt/1e9:.3f} GB/s = {256/dt:.0f} MiB/s", flush=True)
    del res
co = td_corpus.chunk_offsets(n, 320)
t0 = time.perf_counter(); toks, toffs = enc.encode_batch_to_numpy(x, co); dt = time.perf_counter() - t0
out["python_encode_batch_to_numpy"] = {"s": round(dt, 4), "GB/s": round(n / dt / 1e9, 2)}
print(f"Python encode_batch_to_numpy (bytes + offsets -> int32 ids + offsets), 256 MiB: {dt*1e3:.1f} ms = {n/dt/1e9:.2f} GB/s", flush=True)
# ---- one-string calls (the reference's tests/performance_benchmark.py:239-387 shapes) ---------------------------------
samples = {
    "hello": "Hello, world!",
    "sentence": "The quick brown fox jumps over the lazy dog. " * 2,
    "paragraph_900": ("Natural language processing enables computers to understand, interpret and generate human language. " * 9)[:900],
    "code_400": "def fibonacci(n):\n    if n <= 1:\n        return n\n    return fibonacci(n-1) + fibonacci(n-2)\n" * 4,
    "unicode": "Hello 世界! Привет мир! مرحبا بالعالم! 🌍🚀✨ " * 3,
    "text_4k": ("Lorem ipsum dolor sit amet, consectetur adipiscing elit, sed do eiusmod tempor incididunt ut labore. " * 41)[:4000],
    "text_16k": ("Lorem ipsum dolor sit amet, consectetur adipiscing elit, sed do eiusmod tempor incididunt ut labore. " * 164)[:16000],
}
for nm, s in samples.items():
    for _ in range(20): enc.encode(s)
    t0 = time.perf_counter()
    for _ in range(200): enc.encode(s)
    us = (time.perf_counter() - t0) / 200 * 1e6
    ids = enc.encode(s)
    t0 = time.perf_counter()
    for _ in range(200): enc.decode(ids)
    usd = (time.perf_counter() - t0) / 200 * 1e6
    out[f"encode_us_{nm}"] = round(us, 1); out[f"decode_us_{nm}"] = round(usd, 1)
    print(f"enc.encode({nm}: {len(s.encode())} bytes -> {len(ids)} ids): {us:.1f} us per call; decode {usd:.1f} us", flush=True)
print(json.dumps(out))
