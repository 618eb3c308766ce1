"""CHECKER HERE = the heap form of the reference's merge loop (oracle/port.py), not the compiled reference: tiktoken.cpp:322-343 is quadratic
and a single piece of 50 KB takes it seconds, a megabyte hours; the heap form is pinned to the quadratic loop and to the compiled reference
on pieces up to 20 KB (tests/test_oracle.py), and test_grid_path_equals_the_compiled_reference_up_to_20_kb below puts pieces of that size
through the compiled reference itself on the grid path.
td_giant_pieces over all workgroups of the launch (round 5; TD_OPT_GIANT_COOP_MIN): a piece above the limit is
swept by every workgroup together, with grid barriers between the sweeps — same ids as one workgroup per piece and as the heap form of
the reference's merge loop (oracle/port.py, pinned against tiktoken.cpp:322-343 by tests/test_oracle.py).  The limit is turned down
to 1 KiB so that pieces of every shape take the grid path: stretches of a few parts, borders between workgroups inside runs of equal
ranks, a list of pieces longer than the kernel's (GP_LIST_CAP = 1024), pieces of both kinds in one call."""
from __future__ import annotations

import random
import time

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tok():
    from tokendagger_amd import capi
    pat, mr, special = H.llama4()
    t = capi.HipTokenizer(pat, mr, special, device=0)
    yield t
    t.close()


def _pieces(rng):
    letters = "abcdefghijklmnopqrstuvwxyz"
    out = [b"a" * 70_001, b" " * 33_333, ("ab" * 9_000).encode(), ("abc" * 5_001).encode(),
           "".join(rng.choice("ACGT") for _ in range(40_000)).encode(),
           "".join(rng.choice(letters) for _ in range(50_000)).encode(),
           "".join(rng.choice(letters) for _ in range(4097)).encode(),
           "".join(rng.choice(letters) for _ in range(8192 + 1)).encode(),
           "".join(rng.choice("eeeeeeetttttaaaaooooiiinnnssshhrrdlcumwfgypbvkjxqz") for _ in range(30_000)).encode(),
           b"".join(bytes([rng.choice(b"abc")]) * rng.randrange(1, 9) for _ in range(8_000)),
           ("=" * 20_000 + "\n").encode(), ("的" * 5_000).encode("utf-8"), ("é" * 3_000 + "ü" * 3_000).encode("utf-8"),
           b"x" + b"\t" * 5000 + b"y", ("Ab" * 3000 + "cD" * 3000).encode(),
           # runs of equal ranks across the borders of the stretches (4096 parts each at first): a long run, a change, a long run
           b"a" * 4095 + b"b" * 4098 + b"a" * 4096 + b"c", b"z" * 4096 + b"y" * 4096, b"q" * 8191 + b"r",
           "".join(rng.choice("01") for _ in range(20_000)).encode(), b"short doc", b"Hello, world! An ordinary sentence in between."]
    return out


def test_grid_path_equals_the_heap_oracle_and_the_single_workgroup(tok):
    from oracle import port
    from tokendagger_amd import capi
    O = H.port_tokenizer()
    rng = random.Random(5)
    docs = _pieces(rng)
    port.set_heap_threshold(2048)
    try:
        text, offs = H.pack_docs(docs)
        want_t, want_o = O.encode_batch(text, offs)
        got = {}
        for coop_min in (1024, 1 << 30, 16384):  # everything on the grid path | nothing | the default
            tok.set_option(capi.TD_OPT_GIANT_COOP_MIN, coop_min)
            for rep in range(2):
                t0 = time.perf_counter()
                got_t, got_o = tok.encode_batch(text, offs)
                dt = time.perf_counter() - t0
            print(f"giant pieces, limit {coop_min}: {len(text)} bytes in {dt * 1e3:.1f} ms")
            assert np.array_equal(got_o, want_o), coop_min
            assert np.array_equal(got_t, want_t), coop_min
            got[coop_min] = got_t
        assert tok.decode_bytes(got[1024]) == text
        # one by one on the grid path (a call with a single listed piece)
        tok.set_option(capi.TD_OPT_GIANT_COOP_MIN, 1024)
        for d in docs[:12]:
            assert np.array_equal(tok.encode(d), O.encode(d)), d[:16]
    finally:
        port.set_heap_threshold(4096)
        tok.set_option(capi.TD_OPT_GIANT_COOP_MIN, 16384)


def test_grid_path_equals_the_compiled_reference_up_to_20_kb(tok):
    """Pieces of 1.5 .. 20 KB — what the reference's quadratic loop (tiktoken.cpp:322-343) still answers in seconds — through the COMPILED
    REFERENCE, on the grid path (limit 1 KiB) and with a workgroup each."""
    from oracle import ref
    from tokendagger_amd import capi
    assert ref.available(), "oracle/_ref/libtdref.so is missing: build it where /root/reference exists (oracle/build_ref.sh)"
    R = H.ref_tokenizer()
    rng = random.Random(17)
    letters = "abcdefghijklmnopqrstuvwxyz"
    docs = ["".join(rng.choice(letters) for _ in range(20_000)).encode(), b"a" * 12_000, ("ab" * 5_000).encode(), ("xyz" * 2_000 + "q").encode(),
            "".join(rng.choice("ACGT") for _ in range(9_000)).encode(), ("的" * 2_000).encode("utf-8"), b" " * 1_500, b"=" * 4_097,
            "".join(rng.choice("eeeetttaaooinshrdlu") for _ in range(15_000)).encode(), b"an ordinary sentence in between."]
    text, offs = H.pack_docs(docs)
    _, want_t, want_o = R.encode_batch(np.frombuffer(text, dtype=np.uint8), np.asarray(offs, dtype=np.int64), n_threads=8, want_tokens=True)
    try:
        for coop_min in (1024, 1 << 30):
            tok.set_option(capi.TD_OPT_GIANT_COOP_MIN, coop_min)
            got_t, got_o = tok.encode_batch(text, offs)
            assert np.array_equal(got_o, want_o), coop_min
            assert np.array_equal(got_t, want_t), coop_min
    finally:
        tok.set_option(capi.TD_OPT_GIANT_COOP_MIN, 16384)


def test_more_listed_pieces_than_the_list_holds(tok):
    """1100 pieces above the limit in one call: the first 1024 to arrive are swept by the grid, the others by a workgroup each."""
    from oracle import port
    from tokendagger_amd import capi
    O = H.port_tokenizer()
    rng = random.Random(6)
    kinds = [lambda n: b"a" * n, lambda n: ("ab" * n)[:n].encode(), lambda n: "".join(rng.choice("abcdefgh") for _ in range(n)).encode()]
    docs = [kinds[i % 3](1030 + (i * 7) % 300) for i in range(1100)]
    port.set_heap_threshold(1024)
    try:
        text, offs = H.pack_docs(docs)
        want_t, want_o = O.encode_batch(text, offs)
        tok.set_option(capi.TD_OPT_GIANT_COOP_MIN, 1024)
        got_t, got_o = tok.encode_batch(text, offs)
        assert np.array_equal(got_o, want_o) and np.array_equal(got_t, want_t)
    finally:
        port.set_heap_threshold(4096)
        tok.set_option(capi.TD_OPT_GIANT_COOP_MIN, 16384)


def test_a_megabyte_of_random_letters_within_100_ms(tok):
    """VERDICT r4 item 8's mark: 1 MB of random letters <= 100 ms (0.55 s on one workgroup), bit-exact against the heap oracle."""
    from oracle import port
    O = H.port_tokenizer()
    rng = random.Random(77)
    d = "".join(rng.choice("abcdefghijklmnopqrstuvwxyz") for _ in range(1_000_000)).encode()
    port.set_heap_threshold(2048)
    try:
        want = O.encode(d)
    finally:
        port.set_heap_threshold(4096)
    tok.encode(d)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        ids = tok.encode(d)
        best = min(best, time.perf_counter() - t0)
    assert np.array_equal(ids, want)
    print(f"giant pieces: 1 MB of random letters over the grid: {best * 1e3:.1f} ms (host copies included)")
    assert best < 0.1, f"{best * 1e3:.0f} ms"


def test_concurrent_handles_with_listed_pieces_do_not_starve_each_other(tok):
    """Four handles (td_clone), four host threads, each encoding documents with a piece for all workgroups at the same time.  A launch of
    td_giant_pieces takes half of the CUs, so two of them run side by side; further ones cannot get all their workgroups resident — their
    first grid barrier (gp_grid_meet) gives up after 50 ms, for all of the launch's workgroups alike, and each workgroup merges what it had
    listed alone.  Either way: the reference's ids, no TD_E_HIP, no hang."""
    import threading
    from oracle import port
    O = H.port_tokenizer()
    rng = random.Random(9)
    docs = ["".join(rng.choice("abcdefghijklmnopqrstuvwxyz") for _ in range(n)).encode() for n in (120_000, 60_000, 90_000, 40_000)]
    port.set_heap_threshold(2048)
    try:
        want = [O.encode(d) for d in docs]
    finally:
        port.set_heap_threshold(4096)
    handles = [tok] + [tok.clone() for _ in range(3)]
    errors, times = [], [0.0] * 4
    go = threading.Barrier(4)

    def work(k):
        try:
            t = handles[k]
            t.encode(docs[k])  # (workspace)
            go.wait()
            t0 = time.perf_counter()
            for it in range(6):
                d = (k + it) % 4
                ids = t.encode(docs[d])
                if not np.array_equal(ids, want[d]):
                    errors.append((k, it, "ids differ"))
            times[k] = time.perf_counter() - t0
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))
            try:
                go.abort()
            except Exception:  # noqa: BLE001
                pass

    th = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for x in th:
        x.start()
    for x in th:
        x.join(timeout=120)
    assert not any(x.is_alive() for x in th), "a thread hangs"
    for h in handles[1:]:
        h.close()
    assert not errors, errors
    print(f"four handles, six pieces each at the same time: {[round(v * 1e3) for v in times]} ms per thread")
