"""Sixth member of the split-pattern family: Qwen2 / Qwen2.5 / Qwen3 (tokenizer.json pre_tokenizer) = the cl100k_base
pattern with single-digit number pieces.  Pinned like the others: the compiled reference (PCRE2 runs any pattern) against
the restatement's variant on fuzz and golden text; CPU twin and GPU against the restatement."""
import random

import numpy as np
import pytest

import helpers as H
from oracle import port, ref
from tokendagger_amd import vocab_io

EDGE = ["don't", "'sup", "x'ſ", "12'345", "1234567", " 12 345", "a1b22c333", "٣٤٥٦", "１２３", "3.14159", "'ll9", "é9",
        "0\n1\r\n22", "  7  ", "v2.10.3-rc1", "中文123字", " 's9"]


def _strings(n, seed):
    rng = random.Random(seed)
    for i in range(n):
        if i < len(EDGE):
            yield EDGE[i].encode("utf-8")
        elif i % 5 == 0:
            yield "".join(rng.choice("0123456789 a.") for _ in range(rng.randint(1, 40))).encode()
        else:
            yield (H.fuzz_string(rng) if i % 2 else H.random_unicode_string(rng)).encode("utf-8")


@pytest.mark.skipif(not ref.available(), reason="compiled reference (oracle/_ref) not built")
def test_restatement_vs_compiled_reference(golden):
    _, mr, special = H.llama4()
    R = ref.RefTokenizer(vocab_io.QWEN2_PAT_STR, mr, special)
    O = port.OracleTokenizer(mr, port.VARIANT_QWEN2)
    R_cl = H.ref_tokenizer_cl100k()
    assert not np.array_equal(R.split(b"a 123 b"), R_cl.split(b"a 123 b"))  # (not cl100k: digits one by one)
    for s in _strings(4000, 31):
        assert np.array_equal(port.split(s, port.VARIANT_QWEN2), R.split(s)), repr(s)
        assert np.array_equal(O.encode(s), R.encode(s)), repr(s)
    text, offs = golden["text"].tobytes(), golden["offsets"]
    for d in range(0, len(offs) - 1, 7):
        doc = text[offs[d]:offs[d + 1]]
        assert np.array_equal(port.split(doc, port.VARIANT_QWEN2), R.split(doc)), golden["names"][d]
        assert np.array_equal(O.encode(doc), R.encode(doc)), golden["names"][d]


def test_twin_vs_restatement(golden):
    _, mr, special = H.llama4()
    tw = H.Twin(vocab_io.QWEN2_PAT_STR, mr, special)
    O = port.OracleTokenizer(mr, port.VARIANT_QWEN2)
    text, offs = golden["text"].tobytes(), golden["offsets"]
    toks, toffs = tw.encode_batch(text, offs)          # tile windows, whole-word rules (none for this member), chains
    etoks, eoffs = O.encode_batch(text, offs)
    assert np.array_equal(toffs, eoffs) and np.array_equal(toks, etoks)
    bad, n = tw.sync_violations(text, offs)            # the sync rules stay provable for this pattern
    assert bad == 0 and n > 100000
    bad, unres, checked = tw.bits_check(text, offs)    # bit-parallel scanner == byte scanner
    assert bad == 0 and checked > 100000
    bad, st = tw.word_rules_check(text, offs)
    assert bad == 0 and st[0] > 1000
    rng = random.Random(32)
    for it in range(8):
        docs = [s for s in _strings(rng.randint(20, 200), 100 + it)]
        docs += [(rng.choice(["7", "12 ", "a1", " 9"]) * rng.randint(50, 6000)).encode() for _ in range(3)]
        t, o = H.pack_docs(docs)
        toks, toffs = tw.encode_batch(t, o)
        etoks, eoffs = O.encode_batch(t, o)
        assert np.array_equal(toffs, eoffs) and np.array_equal(toks, etoks)
        bad, _ = tw.sync_violations(t, o)
        assert bad == 0


@pytest.mark.gpu
def test_gpu_qwen2_parity(golden):
    import td_corpus
    from tokendagger_amd import capi
    _, mr, special = H.llama4()
    tok = capi.HipTokenizer(vocab_io.QWEN2_PAT_STR, mr, special, device=0)
    O = port.OracleTokenizer(mr, port.VARIANT_QWEN2)
    text, offs = golden["text"], golden["offsets"]
    toks, toffs = tok.encode_batch(text, offs)
    etoks, eoffs = O.encode_batch(text.tobytes(), offs)
    assert np.array_equal(toffs, eoffs) and np.array_equal(toks, etoks)
    rng = random.Random(33)
    for it in range(4):
        docs = [s for s in _strings(400, 200 + it)]
        docs += [(rng.choice(["7", "12 ", "a1", " 9", "it's "]) * rng.randint(100, 9000)).encode() for _ in range(5)]
        t, o = H.pack_docs(docs)
        toks, toffs = tok.encode_batch(t, o)
        etoks, eoffs = O.encode_batch(t, o)
        assert np.array_equal(toffs, eoffs) and np.array_equal(toks, etoks)
    for gen in (td_corpus.mixed, td_corpus.code, td_corpus.english):
        x, o = gen(2 << 20, seed=9)
        toks, toffs = tok.encode_batch(x, o)
        etoks, eoffs = O.encode_batch(x.tobytes(), o)
        assert np.array_equal(toffs, eoffs) and np.array_equal(toks, etoks)
    tok.close()
