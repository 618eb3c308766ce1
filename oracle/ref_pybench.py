"""TEST INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg): the reference's benchmark METHOD on the host cores.

    python oracle/ref_pybench.py <corpus.bin> <n_bytes> <threads> [runs]

Loads the reference's OWN Python extension (oracle/_ref/refmod/_tokendagger_core*.so = /root/reference/src/py_binding.cpp +
tiktoken.cpp, unmodified, built by oracle/build_ref.sh) in a process of its own — its module and class names are the
product's, the two cannot share an interpreter — builds CoreBPE the way the reference's wrapper does
(/root/reference/tokendagger/wrapper.py:86-112: one VocabItem per token, token_bytes as a list of ints; the specials
also entered as regular tokens, tests/throughput_test.py:211-213) and times what the reference's throughput test times
(tests/throughput_test.py:399-422): the text cut into threads x 10 equal character slices, ONE
Tokenizer.encode_batch(chunks, num_threads=threads) call = ThreadPoolExecutor(threads).map(encode) with
encode(text) = CoreBPE.encode(text, set())[0] (wrapper.py:159-196, 212-235), Python lists of ints out.
Prints one JSON line: {"seconds": [...], "bytes": n, "threads": T, "tokens": k}.
"""
from __future__ import annotations

import importlib.util
import json
import sys
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent


def _load(name: str, path: Path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    corpus, n_bytes, threads = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    runs = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    so = next((HERE / "_ref" / "refmod").glob("_tokendagger_core*.so"))
    core = _load("_tokendagger_core", so)
    vocab_io = _load("td_vocab_io_standalone", HERE.parent / "tokendagger_amd" / "vocab_io.py")  # (plain Python: no package import)
    _, pat, ranks, special = vocab_io.load_tdv(vocab_io.default_vocab_path())
    merged = dict(ranks)
    for k, v in special.items():
        merged[k.encode("utf-8")] = v
    items = []
    for tb, r in merged.items():
        it = core.VocabItem()
        it.rank = r
        it.token_bytes = list(tb)
        it.token_string = ""
        items.append(it)
    sitems = []
    for s, r in special.items():
        it = core.VocabItem()
        it.rank = r
        it.token_bytes = list(s.encode("utf-8"))
        it.token_string = s
        sitems.append(it)
    bpe = core.CoreBPE(pat, items, sitems)
    with open(corpus, "rb") as f:
        text = f.read(n_bytes).decode("utf-8")  # (ASCII corpus: characters = bytes, as in the reference's generator)
    n_chunks = threads * 10
    size = len(text) // n_chunks
    chunks = [text[i * size:(i + 1) * size] if i < n_chunks - 1 else text[i * size:] for i in range(n_chunks)]
    empty = set()

    def encode(t):
        return bpe.encode(t, empty)[0]

    def encode_batch():
        with ThreadPoolExecutor(threads) as e:
            return list(e.map(encode, chunks))

    encode_batch()  # warm-up (thread arenas, page faults)
    secs, ntok = [], 0
    for _ in range(runs):
        t0 = time.perf_counter()
        out = encode_batch()
        secs.append(time.perf_counter() - t0)
        ntok = sum(len(o) for o in out)
        del out
    print(json.dumps({"seconds": secs, "bytes": len(text.encode("utf-8")), "threads": threads, "chunks": n_chunks, "tokens": ntok}))


if __name__ == "__main__":
    main()
