"""The oracle itself: plain-C restatement (oracle/td_oracle.c) pinned against the golden vectors that the
compiled reference produced, and — when oracle/_ref is present — against the compiled reference live."""
import random

import numpy as np
import pytest

import cases
import helpers as H
from oracle import port, ref


def _docs(golden):
    text, offs = golden["text"].tobytes(), golden["offsets"]
    for d in range(len(offs) - 1):
        yield d, text[offs[d]:offs[d + 1]]


def test_port_matches_golden_encode(golden):
    O = H.port_tokenizer()
    enc, eo = golden["enc"], golden["enc_offsets"]
    for d, doc in _docs(golden):
        got = O.encode(doc)
        assert np.array_equal(got, enc[eo[d]:eo[d + 1]]), golden["names"][d]


def test_port_matches_golden_split(golden):
    H.port_tokenizer()
    pe, po = golden["piece_ends"], golden["piece_offsets"]
    for d, doc in _docs(golden):
        assert np.array_equal(port.split(doc), pe[po[d]:po[d + 1]]), golden["names"][d]


def test_port_encode_ordinary_same_as_encode_on_golden(golden):
    O = H.port_tokenizer()
    assert golden["ord_same"].all()
    for d, doc in list(_docs(golden))[:400]:
        assert np.array_equal(O.encode_ordinary(doc), O.encode(doc))


def test_port_decode_known_answers(golden):
    O = H.port_tokenizer()
    for ids, exp in zip(golden["decode_ids"], golden["decode_bytes"]):
        assert O.decode_bytes(ids) == exp
    with pytest.raises(port.OracleError):
        O.decode_bytes([999999999])


def test_known_answers_from_survey():
    # sample Llama-4 ids produced by the reference
    O = H.port_tokenizer()
    assert O.encode(b"Hello, world!").tolist() == [19873, 24, 3817, 13]
    assert O.encode(b" ").tolist() == [220]
    assert O.encode(b"\n\t").tolist() == [198, 197]


@pytest.mark.skipif(not ref.available(), reason="compiled reference (oracle/_ref) not built")
def test_port_vs_compiled_reference_fuzz():
    R, O = H.ref_tokenizer(), H.port_tokenizer()
    assert R.pcre2_version()[1].startswith("14."), "class tables were probed from Unicode 14 PCRE2"
    rng = random.Random(11)
    for i in range(6000):
        s = (H.fuzz_string(rng) if i % 2 else H.random_unicode_string(rng)).encode("utf-8")
        assert np.array_equal(port.split(s), R.split(s)), repr(s)
        assert np.array_equal(O.encode(s), R.encode(s)), repr(s)
        if i % 10 == 0:
            assert np.array_equal(O.encode_ordinary(s), R.encode_ordinary(s)), repr(s)


@pytest.mark.skipif(not ref.available(), reason="compiled reference (oracle/_ref) not built")
def test_compiled_reference_still_matches_golden(golden):
    R = H.ref_tokenizer()
    enc, eo = golden["enc"], golden["enc_offsets"]
    for d, doc in list(_docs(golden))[::7]:
        assert np.array_equal(R.encode(doc), enc[eo[d]:eo[d + 1]])


def test_unknown_byte_is_an_error_not_a_garbage_id():
    # toy vocab without 'c': the reference returns a garbage id for the lone byte; oracle raises
    O = port.OracleTokenizer({b"a": 0, b"b": 1, b"ab": 2})
    assert O.encode(b"ab").tolist() == [2]
    with pytest.raises(port.OracleError):
        O.encode(b"c")
    with pytest.raises(port.OracleError):
        O.encode(b"abc")


def test_class_table_spot_checks():
    H.port_tokenizer()
    C = port.class_of_cp
    assert C(0x20) == 3 and C(0x0A) == 5 and C(0x0D) == 5 and C(0x09) == 4
    assert C(0x180E) == 4, "U+180E is \\s under PCRE2 10.39 UCP (unlike Python/Rust)"
    assert C(ord("'")) == 1 and C(ord("/")) == 2
    assert C(ord("A")) == 6 and C(ord("a")) == 7 and C(0x4E2D) == 8 and C(0x0301) == 9 and C(ord("7")) == 10
    assert C(0x01C5) == 6, "Lt counts as upper"
    assert C(0x02B0) == 8, "Lm is in both letter classes"
    assert C(0x1F600) == 0 and C(0x10FFFF) == 0


def test_code_corpus_fixture_matches_the_restatement():
    """BASELINE config 5's file set (tests/golden/code_corpus.npz: the 21 files tests/code_performance_benchmark.py
    selects, ids from the compiled reference): the restatement reproduces every file's ids."""
    g = np.load(H.ROOT / "tests" / "golden" / "code_corpus.npz", allow_pickle=False)
    text, offs, enc, eo = g["text"].tobytes(), g["offsets"], g["enc"], g["enc_offsets"]
    assert len(offs) - 1 == 21 and len(text) == 2146667
    O = H.port_tokenizer()
    for d in range(len(offs) - 1):
        assert np.array_equal(O.encode(text[offs[d]:offs[d + 1]]), enc[eo[d]:eo[d + 1]]), str(g["names"][d])
    import td_corpus
    x, o = td_corpus.code_files(5 << 20)
    assert len(x) == 2 * 2146667 and len(o) - 1 == 42 and x[2146667:].tobytes() == text


def test_heap_form_of_the_merge_loop_is_the_same_function():
    """oracle/td_oracle.c has the reference's quadratic merge loop and an O(n log n) heap form for pieces the quadratic one
    cannot finish (the GPU suite checks megabyte pieces against it): both forms give the same ids, and on pieces the
    compiled reference still finishes (<= 20 KB) they give the reference's."""
    O = H.port_tokenizer()
    rng = random.Random(99)
    docs = [b"a" * 20000, b" " * 9000, ("ab" * 4000).encode(), "".join(rng.choice("ACGT") for _ in range(12000)).encode(),
            "".join(rng.choice("abcdefghijklmnopqrstuvwxyz") for _ in range(9000)).encode(), ("的一是不" * 1200).encode("utf-8"),
            ("Ab" * 2500 + "cD" * 2500).encode(), b"x" + b"\t" * 3000, ("=-" * 3000).encode(), ("aaab" * 2500).encode()]
    docs += ["".join(rng.choice("abAB ") for _ in range(rng.randint(2, 400))).replace(" ", "").encode() or b"ab" for _ in range(300)]
    try:
        for d in docs:
            port.set_heap_threshold(1 << 40)
            quad = O.encode(d)
            port.set_heap_threshold(0)
            heap = O.encode(d)
            assert np.array_equal(quad, heap), d[:40]
        if ref.available():
            R = H.ref_tokenizer()
            port.set_heap_threshold(0)
            for d in docs[:10]:
                assert np.array_equal(O.encode(d), R.encode(d)), d[:40]
    finally:
        port.set_heap_threshold(4096)


@pytest.mark.skipif(not ref.available(), reason="compiled reference (oracle/_ref) not built")
@pytest.mark.parametrize("gen,size", [("english", 8 << 20), ("mixed", 4 << 20), ("code", 4 << 20), ("chat", 2 << 20)])
def test_corpora_of_the_gpu_parity_tests_restatement_equals_reference(gen, size):
    """tests/test_gpu_parity.py::test_corpora_vs_oracle compares the device with the RESTATEMENT on these very corpora (same
    generators, sizes and seed); the restatement and the product read the same generated Unicode class tables, so here the
    restatement is held to the compiled reference (PCRE2's own tables) on the same bytes: document by
    document, and the whole corpus as one document."""
    import td_corpus
    x, offs = getattr(td_corpus, gen)(size, seed=7)
    O = H.port_tokenizer()
    R = H.ref_tokenizer()
    pt, po = O.encode_batch(x.tobytes(), offs)
    _, rt, ro = R.encode_batch(x, offs, n_threads=8, want_tokens=True)
    assert np.array_equal(po, ro) and np.array_equal(pt, rt)
    if gen == "english":
        one = np.asarray([0, len(x)], dtype=np.int64)
        pt, _ = O.encode_batch(x.tobytes(), one)
        _, rt, _ = R.encode_batch(x, one, n_threads=1, want_tokens=True)
        assert np.array_equal(pt, rt)
