"""Runs of digits through the whole-word rules (td_common.h split_unresolved_heads): \\p{N}{1,3} cuts a run every three digits
(/root/reference/src/main.cpp:114, the Llama-4 pattern), and since round 6 the rules place the second piece of a run of four to six ASCII
digits themselves instead of leaving the run to the piece-by-piece matcher.  Checked on the CPU twin of the device algorithm against the
oracle: runs of every length, digits of two and three bytes among them, document starts inside runs, runs across stride and tile ends."""
import random

import numpy as np

import helpers as H

DIGITS = list("0123456789") * 6 + list("٣३５") + ["½", "Ⅷ"]  # (1-, 2- and 3-byte \p{N})
SEPS = [" ", ".", ",", "-", "x", "e", "\n", "", "_", ":", "/", " #", "'s"]


def _docs(rng, n):
    docs = []
    for _ in range(n):
        parts = []
        for _ in range(rng.randint(1, 40)):
            run = "".join(rng.choice(DIGITS) for _ in range(rng.choice([1, 2, 3, 4, 4, 5, 5, 6, 6, 7, 8, 9, 12, 40])))
            parts.append(run + rng.choice(SEPS))
        docs.append("".join(parts))
    return docs


def _check(tw, O, docs):
    text, offs = H.pack_docs([d.encode("utf-8") for d in docs])
    toks, toffs = tw.encode_batch(text, offs)
    etoks, eoffs = O.encode_batch(text, offs)
    assert np.array_equal(toffs, eoffs) and np.array_equal(toks, etoks)
    bad, st = tw.word_rules_check(text, offs)
    assert bad == 0
    return st


def test_digit_runs_llama4_pattern():
    rng = random.Random(5)
    st = _check(H.twin_llama4(), H.port_tokenizer(), _docs(rng, 400))
    assert st[1] < st[0] * 0.5, "most number heads are resolved by the rules"
    # one long document: runs across 32-byte strides and 8 KiB tile ends, at every phase
    for shift in range(0, 40, 7):
        doc = "a" * shift + " ".join("".join(rng.choice("0123456789") for _ in range(rng.choice([4, 5, 6]))) for _ in range(6000))
        _check(H.twin_llama4(), H.port_tokenizer(), [doc, doc[:1000] + "12", "345678"])  # (a document start inside what would be one run)


def test_digit_runs_tekken_pattern():
    rng = random.Random(6)
    _check(H.twin_tekken(), H.port_tokenizer_tekken(), _docs(rng, 300))  # (one digit a piece: the rule does not apply)


def test_digit_runs_cl100k_pattern():
    rng = random.Random(7)
    _check(H.twin_cl100k(), H.port_tokenizer_cl100k(), _docs(rng, 300))
