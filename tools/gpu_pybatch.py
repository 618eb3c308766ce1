"""The unchanged drop-in call: `enc.encode_batch(list[str]) -> list[list[int]]` (reference: tokendagger/wrapper.py:212-235 through
src/py_binding.cpp:25-39) on the reference benchmark's chunking (T x 10 equal slices of 256 MiB, tests/throughput_test.py:399-416)
and on the corpus's own paragraphs (64 MiB), checked against encode_batch_to_numpy.  usage: python tools/gpu_pybatch.py [MiB]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from tokendagger_amd import vocab_io
import bench, td_corpus
import tokendagger as tiktoken
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
name, pat, ranks, special = vocab_io.load_tdv(vocab_io.default_vocab_path())
mr = dict(ranks)
for k, v in special.items():
    mr[k.encode("utf-8")] = v
enc = tiktoken.Encoding(name="llama4", pat_str=pat, mergeable_ranks=mr, special_tokens=special)
n = mb << 20
x, offs = bench.build_corpus("english", n, 1000)
text = x.tobytes().decode("ascii")
for T in (8, 32, 256):
    co = td_corpus.chunk_offsets(n, T * 10)
    chunks = [text[co[i]:co[i + 1]] for i in range(T * 10)]
    enc.encode_batch(chunks[:2], num_threads=T)
    best, res = 1e9, None
    for _ in range(3):
        res = None  # (the previous result is freed OUTSIDE the timed region, as oracle/ref_pybench.py frees the reference's)
        t0 = time.perf_counter(); res = enc.encode_batch(chunks, num_threads=T); best = min(best, time.perf_counter() - t0)
    ntok = sum(len(r) for r in res)
    toks, toffs = enc.encode_batch_to_numpy(x, np.asarray(co, dtype=np.int64))
    ok = ntok == len(toks) and all(res[i] == toks[toffs[i]:toffs[i + 1]].tolist() for i in (0, len(res) // 2, len(res) - 1))
    print(f"Python encode_batch(list[str]) -> list[list[int]], {T*10} slices of {mb} MiB: {best:.3f} s = {n/best/1e9:.3f} GB/s = {mb/best:.0f} MiB/s, "
          f"{ntok} ids, equal to encode_batch_to_numpy on the slices checked: {ok}", flush=True)
    del res
# the corpus's own documents (paragraphs): many short lists
m = min(n, 64 << 20)
k = int(np.searchsorted(offs, m, side="right")) - 1
docs = [text[offs[i]:offs[i + 1]] for i in range(k)]
t0 = time.perf_counter(); res = enc.encode_batch(docs); dt = time.perf_counter() - t0
nb = int(offs[k])
print(f"... {k} paragraphs ({nb >> 20} MiB): {dt:.3f} s = {nb/dt/1e9:.3f} GB/s, {sum(len(r) for r in res)} ids", flush=True)
t0 = time.perf_counter(); one = [enc.encode(d) for d in docs[:20000]]; dt = time.perf_counter() - t0
print(f"... enc.encode() one paragraph at a time, 20000 calls: {dt/20000*1e6:.1f} us per call; equal to the batch: {one == res[:20000]}", flush=True)
