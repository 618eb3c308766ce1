"""lp_merge_batched (round 6; csrc/td_kernels.hip): td_small_encode merges a piece of 65 .. 1024 bytes with every pair of the lowest rank
per round, cut where the sequential order would turn elsewhere (rule + CPU model: tools/sim_rank_batches.py, tests/test_rank_batches_model.py).
Here the kernel itself, through single calls of the C ABI (inputs of at most 4 KiB take the one-launch path), against the COMPILED REFERENCE
(bpe_merge, /root/reference/src/tiktoken/tiktoken.cpp:298-368 — quadratic, but these pieces are at most 1 KiB): repetitive pieces (what
/root/reference/tests/performance_benchmark.py's "repetitive" and "whitespace" cases are), random ones, multi-byte characters, every length
around the lanes' sixteen positions, and toy vocabularies whose ranks are NOT in merge order (where the cut is taken)."""
from __future__ import annotations

import itertools
import random

import numpy as np
import pytest

import helpers as H
from oracle import ref
from tokendagger_amd import capi

pytestmark = pytest.mark.gpu


def _check(tok, R, piece: bytes, what):
    got = tok.encode(piece)
    want = R.encode(piece)
    assert np.array_equal(got, want), f"{what}: {piece[:60]!r} ({len(piece)} bytes): ids differ from the reference"


def test_repetitive_and_random_pieces_llama4():
    assert ref.available(), "oracle/_ref/libtdref.so is missing: build it where /root/reference exists (oracle/build_ref.sh)"
    pat, mr, special = H.llama4()
    R = H.ref_tokenizer()
    tok = capi.HipTokenizer(pat, mr, special, device=0)
    try:
        rng = random.Random(41)
        fixed = [b"a" * 1000, b"abc" * 300, b"-" * 500, b" " * 100, b"\n" * 100, b"ab" * 512, b"a" * 65, b"aab" * 341, b"xyzxyzxy" * 100, b"aaaab" * 200,
                 b"ba" * 33 + b"a" * 900, b"=" * 1024, b"a" * 1024, b"ab" * 500 + b"c" * 24, ("é" * 400).encode(), ("的" * 300).encode(), ("ありがとう" * 60).encode(),
                 b"x" * 1023, b"x" * 1025, b"a" * 4096]
        for p in fixed:
            _check(tok, R, p, "fixed")
        for n in list(range(65, 100)) + list(range(250, 262)) + list(range(1008, 1025)):  # (every length around the lanes' stripes)
            _check(tok, R, b"a" * n, "run of a")
            _check(tok, R, bytes(rng.choice(b"ab") for _ in range(n)), "random ab")
        for it in range(300):
            alpha = rng.choice([b"abcdefghijklmnopqrstuvwxyz", b"ab", b"abc", b"etaoinshr", b"ACGT", b"xyzq", b"aeiou", b"a", b"-=", b"lI"])
            n = rng.randrange(65, 1025)
            piece = (b"".join(bytes([rng.choice(alpha)]) * rng.randrange(1, 40) for _ in range(n // 8 + 1))[:n] if rng.random() < 0.5
                     else bytes(rng.choice(alpha) for _ in range(n)))
            _check(tok, R, piece, "random")
        # several long pieces in one call (a wavefront each), between ordinary text
        doc = b"see " + b"a" * 700 + b" and " + b"xy" * 300 + b" or " + b"-" * 90 + b"\n" + ("的" * 120).encode() + b" end"
        _check(tok, R, doc, "several")
    finally:
        tok.close()


def test_vocabularies_whose_ranks_are_not_in_merge_order():
    assert ref.available(), "oracle/_ref/libtdref.so is missing: build it where /root/reference exists (oracle/build_ref.sh)"
    pat, _, _ = H.llama4()
    for seed in range(6):
        r2 = random.Random(seed)
        extra = [bytes(t) for L in range(2, 6) for t in itertools.product(b"abc", repeat=L) if r2.random() < 0.5]
        r2.shuffle(extra)  # a five-letter token may rank below a two-letter one
        mr2 = {t: i for i, t in enumerate([bytes([c]) for c in range(256)] + extra)}
        R = ref.RefTokenizer(pat, mr2, {})
        tok = capi.HipTokenizer(pat, mr2, {}, device=0)
        try:
            for it in range(60):
                if r2.random() < 0.5:
                    piece = bytes(r2.choice(b"abc") for _ in range(r2.randrange(65, 1025)))
                else:
                    piece = b"".join(bytes([r2.choice(b"abc")]) * r2.randrange(1, 12) for _ in range(r2.randrange(10, 200)))[:1024]
                if len(piece) < 65:
                    piece = piece + b"abc" * 30
                _check(tok, R, piece, f"toy vocabulary {seed}")
        finally:
            tok.close()
