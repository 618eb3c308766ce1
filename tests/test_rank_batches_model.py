"""The rule lp_merge_lds<64> merges by since round 6 (every pair of the lowest rank per round, cut where the sequential order would turn
elsewhere; csrc/td_kernels.hip) as a CPU model (tools/sim_rank_batches.py) against the restatement's byte_pair_encode of the piece as given
(OracleTokenizer.merge_piece, oracle/td_oracle.c: the reference's quadratic loop and its heap form; /root/reference/src/tiktoken/tiktoken.cpp:298-368): pieces of 2 .. 1024 bytes — the lengths the kernel path takes —
over the Llama-4 vocabulary and over toy vocabularies whose ranks are NOT in merge order (a merged token may rank at or below the pair that
made it: exactly the case the cut exists for).  The kernel itself is checked on the GPU (tests/test_gpu_rank_batches.py)."""
import itertools
import random
import sys
from pathlib import Path

import numpy as np
import pytest

import helpers as H
from oracle import port

sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tools"))
import sim_rank_batches as S  # noqa: E402

pytestmark = pytest.mark.skipif(not port.available(), reason="oracle/_build/libtdoracle.so not built")


def _same(ids, want):
    return len(ids) == len(want) and bool((np.asarray(ids, dtype=np.int64) == want).all())


def test_llama4_vocabulary_pieces_of_the_kernels_lengths():
    _, mr, _ = H.llama4()
    bm, O = S.RankBatchMerger(mr), port.OracleTokenizer(mr)
    port.set_heap_threshold(0)
    try:
        rng = random.Random(23)
        fixed = [b"a" * 1000, b"abc" * 300, b"-" * 500, b"ab" * 512, b"a" * 65, b"aab" * 341, b"xyzxyzxy" * 100, b"aaaab" * 200, b"ba" * 33 + b"a" * 900]
        for it in range(260):
            if it < len(fixed):
                piece = fixed[it]
            else:
                alpha = rng.choice([b"abcdefghijklmnopqrstuvwxyz", b"ab", b"abc", b"etaoinshr", b"ACGT", b"xyzq", b"aeiou", b"a", b"-=", b"01"])
                n = rng.randrange(2, 1025)
                piece = (b"".join(bytes([rng.choice(alpha)]) * rng.randrange(1, 40) for _ in range(n // 8 + 1))[:n] if rng.random() < 0.5
                         else bytes(rng.choice(alpha) for _ in range(n)))
            ids, rounds, merges = bm.merge(piece)
            assert _same(ids, O.merge_piece(piece)), piece[:80]
            assert merges == len(piece) - len(ids)
    finally:
        port.set_heap_threshold(4096)


def test_multi_byte_characters():
    _, mr, _ = H.llama4()
    bm, O = S.RankBatchMerger(mr), port.OracleTokenizer(mr)
    rng = random.Random(3)
    for it in range(60):
        s = "".join(rng.choice("的一是不了人我在有他这为之大来以个中上们到说国和地也子时道出而要于就下得可你年生あいうえおカタカナ한국어éßü") * rng.randrange(1, 6) for _ in range(rng.randrange(1, 120)))
        piece = s.encode("utf-8")[:1024]
        # (cut at a character boundary)
        while piece and (piece[-1] & 0xC0) == 0x80:
            piece = piece[:-1]
        if len(piece) > 1 and (piece[-1] & 0xC0) == 0xC0:
            piece = piece[:-1]
        if len(piece) < 2:
            continue
        ids, _, _ = bm.merge(piece)
        assert _same(ids, O.merge_piece(piece)), s[:40]  # (byte_pair_encode of the bytes as ONE piece: no pre-tokenizer)


def test_vocabularies_whose_ranks_are_not_in_merge_order():
    port.set_heap_threshold(0)
    try:
        for seed in range(16):
            r2 = random.Random(seed)
            extra = [bytes(t) for L in range(2, 6) for t in itertools.product(b"abc", repeat=L) if r2.random() < 0.5]
            r2.shuffle(extra)  # a five-letter token may rank below a two-letter one
            mr2 = {t: i for i, t in enumerate([bytes([c]) for c in range(256)] + extra)}
            bm, O = S.RankBatchMerger(mr2), port.OracleTokenizer(mr2)
            for it in range(40):
                if r2.random() < 0.5:
                    piece = bytes(r2.choice(b"abc") for _ in range(r2.randrange(2, 400)))
                else:
                    piece = b"".join(bytes([r2.choice(b"abc")]) * r2.randrange(1, 12) for _ in range(r2.randrange(1, 60)))
                if len(piece) < 2:
                    continue
                ids, _, _ = bm.merge(piece)
                assert _same(ids, O.merge_piece(piece)), (seed, piece[:80])
    finally:
        port.set_heap_threshold(4096)
