"""Character seeds (tokendagger_amd/csrc/td_common.h, round 5): multi-byte characters whose bytes merge to one token enter the merge
loop as ONE part where the table builder (td_tables.cpp: build_char_seeds) and the neighbour-byte sets prove that this cannot change
the result.  The proof is in td_common.h; this is the model test in front of the GPU ("any shortcut needs a CPU model
test"): the CPU twin runs the very functions the kernel runs (cseed_part_at) and merges from the seeded parts with the reference's
rule (lowest rank, leftmost: /root/reference/src/tiktoken/tiktoken.cpp:322-343); the ids must equal the restatement's — which is held
to the compiled reference in test_oracle.py — on pieces of every script, on random Unicode, on ill-formed UTF-8 and on pieces built
to hit the neighbour-byte sets (a space, a continuation byte, a lead byte next to every seedable character)."""
from __future__ import annotations

import random

import numpy as np
import pytest

import helpers as H
import td_corpus


@pytest.fixture(scope="module")
def twin():
    pat, mr, special = H.llama4()
    return H.Twin(pat, mr, special)


def _check(twin, pieces, what):
    """Every string is ONE piece here, whatever the split pattern would make of it: the merge loop takes any bytes."""
    O = H.port_tokenizer()
    pieces = [p for p in pieces if 0 < len(p) <= 4000]
    got, seeded, parts0 = twin.seeded_merge(pieces)
    for p, g in zip(pieces, got):
        want = O.merge_piece(p)
        assert np.array_equal(g, want), f"{what}: seeded merge differs from the reference's loop on {p!r}: {g.tolist()} != {list(want)}"
    return seeded, parts0, sum(len(p) for p in pieces)


def test_the_llama4_vocabulary_has_seeds(twin):
    assert twin.info(6) > 3000, "most of the one-character tokens of 2..3 bytes may be entered whole"


def test_scripts_and_random_unicode(twin):
    rng = random.Random(11)
    x, o = td_corpus.mixed(1 << 20, seed=3)
    words = [w for w in x.tobytes().split(b" ") if w]
    # whole CJK / Devanagari / Arabic / Cyrillic runs as they come, with and without the blank in front
    pieces = words[:30000] + [b" " + w for w in words[:30000]]
    seeded, parts0, nbytes = _check(twin, pieces, "mixed-script words")
    assert seeded > 20000 and parts0 < nbytes, "characters were entered whole"
    rnd = [H.random_unicode_string(rng, rng.randint(1, 60)).encode("utf-8") for _ in range(20000)]
    _check(twin, rnd, "random unicode")
    cjk = ["".join(chr(rng.randint(0x4E00, 0x9FFF)) for _ in range(rng.randint(1, 50))).encode() for _ in range(5000)]
    kana = ["".join(chr(rng.choice([rng.randint(0x3040, 0x30FF), rng.randint(0x4E00, 0x4F00)])) for _ in range(rng.randint(1, 50))).encode() for _ in range(5000)]
    hangul = ["".join(chr(rng.randint(0xAC00, 0xD7A3)) for _ in range(rng.randint(1, 40))).encode() for _ in range(3000)]
    indic = ["".join(chr(rng.randint(0x0900, 0x0DFF)) for _ in range(rng.randint(1, 40))).encode() for _ in range(5000)]
    thai = ["".join(chr(rng.randint(0x0E00, 0x0E7F)) for _ in range(rng.randint(1, 40))).encode() for _ in range(3000)]
    latin = ["".join(chr(rng.choice([rng.randint(0x61, 0x7A), rng.randint(0xC0, 0x24F), rng.randint(0x370, 0x4FF)])) for _ in range(rng.randint(1, 30))).encode()
             for _ in range(8000)]
    for what, ps in (("CJK", cjk), ("kana + CJK", kana), ("hangul", hangul), ("indic", indic), ("thai", thai), ("latin, greek, cyrillic", latin)):
        _check(twin, ps, what)


def test_every_seedable_character_with_every_kind_of_neighbour(twin):
    """A seedable character between every kind of byte: ASCII letters / blanks / punctuation, continuation bytes of every value (the end
    of another character), lead bytes, nothing."""
    pat, mr, special = H.llama4()
    chars = [b for b in mr if 2 <= len(b) <= 3 and _one_char(b)]
    assert len(chars) > 5000
    rng = random.Random(5)
    before = [b"", b" ", b"a", b"Z", b".", b"\n", b"1"] + [bytes([0xE4, 0xB8, c]) for c in range(0x80, 0xC0)] + [bytes([0xC3, c]) for c in range(0x80, 0xC0, 3)]
    after = [b"", b" ", b"a", b".", "中".encode(), "の".encode(), "다".encode(), "é".encode(), "я".encode(), "न".encode(), "ก".encode(), "\U0001F600".encode()]
    pieces = []
    for c in chars:
        for _ in range(6):
            pieces.append(rng.choice(before) + c + rng.choice(after))
        pieces.append(c + c)
        pieces.append(rng.choice(chars) + c + rng.choice(chars))
    _check(twin, pieces, "characters with neighbours")


def test_ill_formed_utf8(twin):
    """The reference takes any bytes (PCRE2_NO_UTF_CHECK, tiktoken.cpp:91); so do the seeds: truncated characters, stray continuation
    bytes, overlong forms and lead bytes in a row must come out as the loop on bytes has them."""
    rng = random.Random(9)
    alphabet = [0x20, 0x61, 0x80, 0x81, 0xA0, 0xBF, 0xC0, 0xC2, 0xC3, 0xD0, 0xE0, 0xE3, 0xE4, 0xE5, 0xB8, 0xAD, 0x96, 0x87, 0xED, 0xEF, 0xF0, 0x9F, 0x98, 0xFF]
    pieces = [bytes(rng.choice(alphabet) for _ in range(rng.randint(1, 24))) for _ in range(60000)]
    _check(twin, pieces, "ill-formed bytes")


def _one_char(b: bytes) -> bool:
    try:
        return len(b.decode("utf-8")) == 1
    except UnicodeDecodeError:
        return False
