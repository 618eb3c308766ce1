"""The rule td_giant_pieces merges by (bands of ranks per round, csrc/td_kernels.hip) as a CPU model (tools/sim_giant_bands.py)
against the heap form of the reference's merge loop (oracle/td_oracle.c; /root/reference/src/tiktoken/tiktoken.cpp:298-368): random
pieces over the Llama-4 vocabulary and over toy vocabularies whose ranks are NOT in merge order (a merged token may rank below
its parts: the case the bound's second condition exists for).  The kernel itself is checked on the GPU
(tests/test_gpu_parity.py::test_giant_pieces_are_not_quadratic)."""
import itertools
import random
import sys
from pathlib import Path

import numpy as np
import pytest

import helpers as H
from oracle import port

sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tools"))
import sim_giant_bands as G  # noqa: E402

pytestmark = pytest.mark.skipif(not port.available(), reason="oracle/_build/libtdoracle.so not built")


def _same(ids, want):
    return len(ids) == len(want) and bool((np.asarray(ids, dtype=np.int64) == want).all())


def test_llama4_vocabulary_random_letter_pieces():
    _, mr, _ = H.llama4()
    bm, O = G.BandMerger(mr), port.OracleTokenizer(mr)
    port.set_heap_threshold(0)
    try:
        rng = random.Random(11)
        for it in range(120):
            alpha = rng.choice([b"abcdefghijklmnopqrstuvwxyz", b"ab", b"abc", b"etaoinshr", b"ACGT", b"xyzq", b"aeiou"])
            n = rng.randrange(2, 900)
            piece = (b"".join(bytes([rng.choice(alpha)]) * rng.randrange(1, 12) for _ in range(n // 4 + 1)) if rng.random() < 0.3
                     else bytes(rng.choice(alpha) for _ in range(n)))
            ids, rounds, _, _, _ = bm.merge(piece)
            assert _same(ids, O.encode_ordinary(piece)), piece[:80]
            assert rounds <= 40 + len(piece) // 50  # (bands: nowhere near one round per distinct rank)
    finally:
        port.set_heap_threshold(4096)


def test_vocabularies_whose_ranks_are_not_in_merge_order():
    port.set_heap_threshold(0)
    try:
        for seed in range(12):
            r2 = random.Random(seed)
            extra = [bytes(t) for L in range(2, 6) for t in itertools.product(b"abc", repeat=L) if r2.random() < 0.5]
            r2.shuffle(extra)  # a five-letter token may rank below a two-letter one
            mr2 = {t: i for i, t in enumerate([bytes([c]) for c in range(256)] + extra)}
            bm, O = G.BandMerger(mr2), port.OracleTokenizer(mr2)
            for it in range(25):
                piece = bytes(r2.choice(b"abc") for _ in range(r2.randrange(2, 300)))
                ids, _, _, _, _ = bm.merge(piece)
                assert _same(ids, O.encode_ordinary(piece)), (seed, piece[:80])
    finally:
        port.set_heap_threshold(4096)
