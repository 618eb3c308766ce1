#!/usr/bin/env python
"""tests/golden/fullsize_hashes.npz: what the compiled reference (oracle/_ref = the unmodified /root/reference/src/tiktoken/tiktoken.cpp)
gives for the FULL-SIZE corpora of tests/test_gpu_fullsize.py, as one 64-bit hash per 2^18 ids (1 MiB of int32) + one per 2^16 document
offsets — so that the full-size GPU tests have a reference-derived check on a box where oracle/_ref/libtdref.so is not present
(parity evidence must not be skippable).  The corpora are the seeded generators of td_corpus.py through bench.build_corpus.
Run where /root/reference exists:  python tools/make_fullsize_golden.py"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import helpers as H  # noqa: E402

ID_BLOCK, OFF_BLOCK = 1 << 18, 1 << 16
CASES = [("english", 256), ("mixed", 64), ("code", 64), ("english", 1024)]


def block_hashes(a: np.ndarray, block: int) -> np.ndarray:
    a = np.ascontiguousarray(a)
    out = np.zeros((len(a) + block - 1) // block, dtype=np.uint64)
    for i in range(len(out)):
        out[i] = int.from_bytes(hashlib.blake2b(a[i * block:(i + 1) * block].tobytes(), digest_size=8).digest(), "little")
    return out


def main():
    R = H.ref_tokenizer()
    out = {}
    for kind, mb in CASES:
        x, offs = bench.build_corpus(kind, mb << 20, 1000)
        _, et, eo = R.encode_batch(x, offs, n_threads=os.cpu_count() or 1, want_tokens=True)
        key = f"{kind}_{mb}"
        out[key + "_ids"] = block_hashes(et.astype(np.int32), ID_BLOCK)
        out[key + "_offs"] = block_hashes(eo.astype(np.int64), OFF_BLOCK)
        out[key + "_meta"] = np.asarray([len(x), len(offs) - 1, len(et), int(hashlib.blake2b(x.tobytes(), digest_size=8).hexdigest(), 16) >> 1], dtype=np.int64)
        print(key, "bytes", len(x), "docs", len(offs) - 1, "ids", len(et), "id blocks", len(out[key + "_ids"]), flush=True)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "fullsize_hashes.npz"), **out)


if __name__ == "__main__":
    main()
