"""Second member of the split-pattern family: the Mistral tekken pattern (no contraction suffix, single-digit number
pieces).  The tekken.json vocabulary is absent from the reference checkout, so the pattern is pinned on a labelled
surrogate: tekken pattern + Llama-4 vocabulary, golden vectors from the compiled reference (PCRE2 runs any pattern)."""
import random

import numpy as np
import pytest

import helpers as H
from oracle import port, ref


def _docs(golden, step=1):
    text, offs = golden["text"].tobytes(), golden["offsets"]
    for d in range(0, len(offs) - 1, step):
        yield d, text[offs[d]:offs[d + 1]]


def test_pattern_string_is_the_fixture_pattern(tekken_golden):
    assert str(tekken_golden["pattern"]) == H.TEKKEN_PAT
    assert H.TEKKEN_PAT == H.llama4()[0].replace("(?i:'s|'t|'re|'ve|'m|'ll|'d)?", "").replace(r"\p{N}{1,3}", r"\p{N}")


def test_restatement_matches_golden(golden, tekken_golden):
    O = H.port_tokenizer_tekken()
    enc, eo = tekken_golden["enc"], tekken_golden["enc_offsets"]
    pe, po = tekken_golden["piece_ends"], tekken_golden["piece_offsets"]
    differs = 0
    for d, doc in _docs(golden):
        assert np.array_equal(O.encode(doc), enc[eo[d]:eo[d + 1]]), golden["names"][d]
        assert np.array_equal(port.split(doc, port.VARIANT_TEKKEN), pe[po[d]:po[d + 1]]), golden["names"][d]
        differs += not np.array_equal(port.split(doc), pe[po[d]:po[d + 1]])
    assert differs > 200, "the two patterns split many of the fixture documents differently"


@pytest.mark.skipif(not ref.available(), reason="compiled reference (oracle/_ref) not built")
def test_restatement_vs_compiled_reference_fuzz():
    R, O = H.ref_tokenizer_tekken(), H.port_tokenizer_tekken()
    rng = random.Random(12)
    for i in range(4000):
        s = (H.fuzz_string(rng) if i % 2 else H.random_unicode_string(rng)).encode("utf-8")
        assert np.array_equal(port.split(s, port.VARIANT_TEKKEN), R.split(s)), repr(s)
        assert np.array_equal(O.encode(s), R.encode(s)), repr(s)


def test_twin_scanners_match_golden(golden, tekken_golden):
    tw = H.twin_tekken()
    text, offs = golden["text"].tobytes(), golden["offsets"]
    pe, po = tekken_golden["piece_ends"], tekken_golden["piece_offsets"]
    exp = []
    for d in range(len(offs) - 1):
        ends = pe[po[d]:po[d + 1]]
        if len(ends):
            starts = np.concatenate([[0], ends[:-1]])
            if d % 9 == 0:
                assert np.array_equal(tw.split_serial(text[offs[d]:offs[d + 1]]), starts), golden["names"][d]
            exp.append(starts + offs[d])
    got, _ = tw.split_tiled(text, offs)            # tile windows + speculative lanes + slow path
    assert np.array_equal(got, np.concatenate(exp))
    bad, n = tw.sync_violations(text, offs)        # the sync rules stay provable for this pattern
    assert bad == 0 and n > 100000
    bad, unres, checked = tw.bits_check(text, offs)  # bit-parallel scanner == byte scanner
    assert bad == 0 and checked > 100000
    bad, st = tw.word_rules_check(text, offs)         # whole-word boundary rules never resolve a head wrongly
    assert bad == 0 and st[0] > 1000
    bad, checked = tw.arrmask_check(text, offs)
    assert bad == 0 and checked > 100000
    toks, toffs = tw.encode_batch(text, offs)
    assert np.array_equal(toffs, tekken_golden["enc_offsets"]) and np.array_equal(toks, tekken_golden["enc"])


def test_twin_fuzz_vs_restatement():
    tw, O = H.twin_tekken(), H.port_tokenizer_tekken()
    rng = random.Random(77)
    for _ in range(12):
        docs = []
        for _ in range(rng.randint(1, 150)):
            r = rng.random()
            if r < 0.1:
                docs.append((rng.choice(["a", " ", "=", "1", "\n", "A", "x'S", "9'll"]) * rng.randint(50, 6000)).encode())
            else:
                docs.append("".join(H.fuzz_string(rng) for _ in range(rng.randint(1, 30))).encode("utf-8"))
        text, offs = H.pack_docs(docs)
        toks, toffs = tw.encode_batch(text, offs)
        etoks, eoffs = O.encode_batch(text, offs)
        assert np.array_equal(toffs, eoffs) and np.array_equal(toks, etoks)
        bad, _ = tw.sync_violations(text, offs)
        assert bad == 0


@pytest.mark.gpu
def test_gpu_tekken_style_parity(golden, tekken_golden):
    import td_corpus
    from tokendagger_amd import capi
    _, mr, special = H.llama4()
    tok = capi.HipTokenizer(H.TEKKEN_PAT, mr, special, device=0)
    text, offs = golden["text"], golden["offsets"]
    for mode in (0, 1):
        toks, toffs = tok.encode_batch(text, offs, mode=mode)
        assert np.array_equal(toffs, tekken_golden["enc_offsets"]) and np.array_equal(toks, tekken_golden["enc"])
    O = H.port_tokenizer_tekken()
    rng = random.Random(21)
    for _ in range(6):
        docs = ["".join(H.fuzz_string(rng) for _ in range(rng.randint(1, 40))).encode("utf-8") for _ in range(300)]
        docs += [(rng.choice(["7", "it's ", "A1b2 ", "\n"]) * rng.randint(100, 5000)).encode() for _ in range(4)]
        t, o = H.pack_docs(docs)
        toks, toffs = tok.encode_batch(t, o)
        etoks, eoffs = O.encode_batch(t, o)
        assert np.array_equal(toffs, eoffs) and np.array_equal(toks, etoks)
    for gen in (td_corpus.mixed, td_corpus.code, td_corpus.english):
        x, o = gen(2 << 20, seed=5)
        toks, toffs = tok.encode_batch(x, o)
        etoks, eoffs = O.encode_batch(x.tobytes(), o)
        assert np.array_equal(toffs, eoffs) and np.array_equal(toks, etoks)
    # 64 MiB: token count equals the restatement's on a sample, decode round-trips the whole text
    x, o = td_corpus.mixed(64 << 20, seed=6)
    toks, toffs = tok.encode_batch(x, o)
    assert tok.decode_bytes(toks) == x.tobytes()
    d = len(o) // 2
    assert np.array_equal(toks[toffs[d]:toffs[d + 50]], O.encode_batch(x.tobytes(), o[d:d + 51])[0])
