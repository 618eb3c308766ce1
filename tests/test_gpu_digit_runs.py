"""Long runs of digits (round 5).  Under \\p{N}{1,3} a piece start inside a digit run is the run's start + 3 k: no position of the run
is a synchronisation point, so its tiles form a chain that ONE wavefront of td_split_far_tiles walks.  Piece by piece that was ~2 us
per piece (20 KB of digits: 15 ms, a megabyte 0.75 s); FarScan::digit_run_skip now searches the end of the run of ASCII digits 256 bytes
per step and marks the starts in between a word of START bits per lane.  Checked against the compiled reference (PCRE2 on the
reference's own pattern string) for the Llama-4 pattern (groups of three), Qwen2's (digits one by one) and GPT-2's (a run is one piece),
with what cuts a run in every residue mod 3: other bytes, document starts, digits that are not ASCII."""
from __future__ import annotations

import random
import time

import numpy as np
import pytest

import helpers as H
from oracle import ref
from tokendagger_amd import vocab_io

pytestmark = pytest.mark.gpu


def _docs(rng):
    D = "0123456789"
    run = lambda n: "".join(rng.choice(D) for _ in range(n))  # noqa: E731
    docs = []
    for n in (1, 2, 3, 4, 100, 4095, 4096, 4097, 8191, 8192, 8193, 8194, 8195, 20_000, 65_536 + 1, 300_000):
        docs.append(run(n).encode())
    for lead in ("", "a", " ", "ab ", "x" * 8190, "x" * 8191, "x" * 8192, "\n\n"):  # the run starts at every offset mod 3 and around a tile border
        for n in (9000, 9001, 9002, 30_001):
            docs.append((lead + run(n) + " tail 12 345").encode())
    # runs cut by one other byte, every residue; by digits of other scripts (still \p{N}: the groups of three go on across them)
    docs.append("".join(run(rng.randrange(1, 40)) + rng.choice(" .,-a\n") for _ in range(3000)).encode())
    docs.append("".join(run(rng.randrange(8000, 9000)) + rng.choice(["٣", "１２", "४५६", "²", "Ⅷ"]) for _ in range(12)).encode("utf-8"))
    docs.append(("１２３" * 4000 + run(10_000) + "٣" * 3001 + run(10_001)).encode("utf-8"))
    docs.append((run(10_000) + "٣" + run(10_000) + "٣٣" + run(10_000)).encode("utf-8"))
    docs.append(("7" * 8192 + "x" + "7" * 8193 + "y" + "7" * 8194).encode())
    docs.append(("3.14159" + run(50_000) + "e+" + run(5000)).encode())
    return docs


def _check(tok, R, docs):
    text, offs = H.pack_docs(docs)  # (documents back to back: a run that ends a document is followed by the next one's digits)
    got_t, got_o = tok.encode_batch(text, offs)
    want = [R.encode(d) for d in docs]
    want_o = np.concatenate([[0], np.cumsum([len(w) for w in want])])
    assert np.array_equal(got_o, want_o)
    assert np.array_equal(got_t, np.concatenate(want))
    assert tok.decode_bytes(got_t) == bytes(text)


@pytest.mark.skipif(not ref.available(), reason="compiled reference (oracle/_ref) not built")
@pytest.mark.parametrize("pattern", ["llama4", "qwen2", "gpt2", "cl100k"])
def test_digit_runs_equal_the_compiled_reference(pattern):
    from tokendagger_amd import capi
    pat0, mr, special = H.llama4()
    pat = {"llama4": pat0, "qwen2": vocab_io.QWEN2_PAT_STR, "gpt2": vocab_io.GPT2_PAT_STR, "cl100k": vocab_io.CL100K_PAT_STR}[pattern]
    R = ref.RefTokenizer(pat, mr, special)
    tok = capi.HipTokenizer(pat, mr, special, device=0)
    try:
        docs = _docs(random.Random(11))
        if pattern == "gpt2":  # (a run is ONE piece there, and the reference's merge loop is quadratic in its length, tiktoken.cpp:322-343)
            docs = [d for d in docs if len(d) <= 10_000]
        _check(tok, R, docs)
        _check(tok, R, docs[-9:] + docs[:20])
        for d in docs[10:16]:  # one by one too (a single document, other tile phases)
            assert np.array_equal(tok.encode(d), R.encode(d))
    finally:
        tok.close()


def test_a_megabyte_of_digits_is_not_a_second():
    from tokendagger_amd import capi
    pat, mr, special = H.llama4()
    tok = capi.HipTokenizer(pat, mr, special, device=0)
    try:
        rng = random.Random(12)
        d = "".join(rng.choice("0123456789") for _ in range(1_000_000)).encode()
        ids = tok.encode(d)
        t0 = time.perf_counter()
        ids = tok.encode(d)
        dt = time.perf_counter() - t0
        assert len(ids) == (len(d) + 2) // 3 and tok.decode_bytes(ids) == d
        print(f"a megabyte of digits: {dt * 1e3:.1f} ms")
        assert dt < 0.05, f"{dt * 1e3:.0f} ms"
    finally:
        tok.close()
