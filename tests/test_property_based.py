"""Property-based checks (hypothesis) of the host logic against the oracle restatement: arbitrary Unicode text, in
arbitrary document batches, through the CPU twin of the device algorithm (tile windows, speculative lanes, bit-parallel
scanners, table probes, merge) -- for all three members of the split-pattern family."""
import numpy as np
from hypothesis import HealthCheck, given, settings, strategies as st

import helpers as H

# characters that stress the class table: ASCII, Latin-1, combining marks, CJK, modifier letters, digits of other
# scripts, odd whitespace (U+0085, U+180E, U+2028, U+3000, U+FEFF), emoji, the contraction letters
_ALPHA = st.one_of(
    st.sampled_from(list(" \n\r\t'/sStTmMdDlLvVrReE.,!?-_=()[]{}<>|\"#%&*+:;@^~`\\0123456789ſ")),
    st.characters(min_codepoint=0x20, max_codepoint=0x24F),
    st.characters(min_codepoint=0x2B0, max_codepoint=0x36F),
    st.characters(min_codepoint=0x370, max_codepoint=0x6FF),
    st.characters(min_codepoint=0x900, max_codepoint=0x97F),
    st.characters(min_codepoint=0x2000, max_codepoint=0x206F),
    st.characters(min_codepoint=0x3000, max_codepoint=0x30FF),
    st.characters(min_codepoint=0x4E00, max_codepoint=0x4E7F),
    st.characters(min_codepoint=0x1F600, max_codepoint=0x1F64F),
    st.sampled_from(["\u0085", "\u180e", "\u2028", "\u3000", "\u00a0", "\ufeff", "\U000e0001"]),
)
_TEXT = st.text(alphabet=_ALPHA, min_size=0, max_size=120)
_RUN = st.builds(lambda t, k: t * k, st.text(alphabet=_ALPHA, min_size=1, max_size=6), st.integers(1, 900))
_DOCS = st.lists(st.one_of(_TEXT, _RUN), min_size=1, max_size=12)

_CFG = dict(max_examples=300, deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.data_too_large])


def _check(tw, O, docs):
    text, offs = H.pack_docs([d.encode("utf-8") for d in docs])
    toks, toffs = tw.encode_batch(text, offs)
    etoks, eoffs = O.encode_batch(text, offs)
    assert np.array_equal(toffs, eoffs), docs
    assert np.array_equal(toks, etoks), docs
    bad, _ = tw.sync_violations(text, offs)
    assert bad == 0, docs
    if len(text):
        bad, _, _ = tw.bits_check(text, offs)
        assert bad == 0, docs
        bad, _ = tw.word_rules_check(text, offs)  # a head the whole-word rules resolve is exactly one piece
        assert bad == 0, docs


@settings(**_CFG)
@given(_DOCS)
def test_llama4_pattern(docs):
    _check(H.twin_llama4(), H.port_tokenizer(), docs)


@settings(**_CFG)
@given(_DOCS)
def test_tekken_pattern(docs):
    _check(H.twin_tekken(), H.port_tokenizer_tekken(), docs)


@settings(**_CFG)
@given(_DOCS)
def test_cl100k_pattern(docs):
    _check(H.twin_cl100k(), H.port_tokenizer_cl100k(), docs)


@settings(**_CFG)
@given(_DOCS)
def test_gpt2_pattern(docs):
    _check(H.twin_gpt2(), H.port_tokenizer_gpt2(), docs)
