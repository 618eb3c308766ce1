"""Third member of the split-pattern family: cl100k_base / Llama-3 (contraction as an alternative of its own, plain
\\p{L}+ letters, marks are punctuation, no '/' trailer).  Pinned like the tekken member: compiled reference (PCRE2 runs
any pattern) -> golden vectors over the Llama-4 vocabulary; restatement, CPU twin and the GPU checked against them."""
import random

import numpy as np
import pytest

import helpers as H
from oracle import port, ref
from tokendagger_amd import vocab_io


def _docs(golden, step=1):
    text, offs = golden["text"].tobytes(), golden["offsets"]
    for d in range(0, len(offs) - 1, step):
        yield d, text[offs[d]:offs[d + 1]]


EDGE = ["don't", "'sup", "'S", "x'ſ", "''ll", "'", "a'", "éx", "́́", "a/\n/b", "A/\r\n", " 'tis", "\n'd", "'re'RE're",
        "ʰʰ'm", "12'345", "中'文's", "_'t", " 's", "'ſſ"]


def test_pattern_strings(cl100k_golden):
    assert str(cl100k_golden["pattern"]) == H.CL100K_PAT == vocab_io.CL100K_PAT_STR


def test_restatement_matches_golden_and_edges(golden, cl100k_golden):
    O = H.port_tokenizer_cl100k()
    enc, eo = cl100k_golden["enc"], cl100k_golden["enc_offsets"]
    pe, po = cl100k_golden["piece_ends"], cl100k_golden["piece_offsets"]
    differs = 0
    for d, doc in _docs(golden):
        assert np.array_equal(O.encode(doc), enc[eo[d]:eo[d + 1]]), golden["names"][d]
        assert np.array_equal(port.split(doc, port.VARIANT_CL100K), pe[po[d]:po[d + 1]]), golden["names"][d]
        differs += not np.array_equal(port.split(doc), pe[po[d]:po[d + 1]])
    assert differs > 200


@pytest.mark.skipif(not ref.available(), reason="compiled reference (oracle/_ref) not built")
def test_restatement_vs_compiled_reference_fuzz_and_possessive_spelling():
    R, O = H.ref_tokenizer_cl100k(), H.port_tokenizer_cl100k()
    _, mr, special = H.llama4()
    R2 = ref.RefTokenizer(vocab_io.CL100K_PAT_STR_POSSESSIVE, mr, special)  # tiktoken's later spelling: same language
    # the spelling current tiktoken releases ship puts `\s++$` in front of `\s*[\r\n]`: NOT the same language (a trailing
    # whitespace run that contains CR/LF is one piece) -> a variant of its own in the restatement and a scanner flag
    R3 = ref.RefTokenizer(vocab_io.CL100K_PAT_STR_CURRENT, mr, special)
    O3 = port.OracleTokenizer(mr, port.VARIANT_CL100K_EOS)
    tw3 = H.Twin(vocab_io.CL100K_PAT_STR_CURRENT, {bytes([i]): i for i in range(256)})
    assert H.Twin(vocab_io.CL100K_PAT_STR_POSSESSIVE, {bytes([i]): i for i in range(256)}).info(3) == 255
    assert not np.array_equal(R3.split(b"ab\r\t"), R.split(b"ab\r\t"))
    rng = random.Random(13)
    for i in range(4000):
        s = (EDGE[i] if i < len(EDGE) else H.fuzz_string(rng) if i % 2 else H.random_unicode_string(rng)).encode("utf-8")
        want = R.split(s)
        assert np.array_equal(port.split(s, port.VARIANT_CL100K), want), repr(s)
        assert np.array_equal(R2.split(s), want), repr(s)
        want3 = R3.split(s)
        assert np.array_equal(port.split(s, port.VARIANT_CL100K_EOS), want3), repr(s)
        if len(s):
            assert np.array_equal(tw3.split_serial(s), np.concatenate([[0], want3[:-1]])), repr(s)
        assert np.array_equal(O3.encode(s), R3.encode(s)), repr(s)
        assert np.array_equal(O.encode(s), R.encode(s)), repr(s)


def test_twin_scanners_match_golden(golden, cl100k_golden):
    tw = H.twin_cl100k()
    text, offs = golden["text"].tobytes(), golden["offsets"]
    pe, po = cl100k_golden["piece_ends"], cl100k_golden["piece_offsets"]
    exp = []
    for d in range(len(offs) - 1):
        ends = pe[po[d]:po[d + 1]]
        if len(ends):
            starts = np.concatenate([[0], ends[:-1]])
            if d % 9 == 0:
                assert np.array_equal(tw.split_serial(text[offs[d]:offs[d + 1]]), starts), golden["names"][d]
            exp.append(starts + offs[d])
    got, _ = tw.split_tiled(text, offs)
    assert np.array_equal(got, np.concatenate(exp))
    bad, n = tw.sync_violations(text, offs)
    assert bad == 0 and n > 100000
    bad, unres, checked = tw.bits_check(text, offs)
    assert bad == 0 and checked > 100000
    bad, st = tw.word_rules_check(text, offs)         # whole-word boundary rules never resolve a head wrongly
    assert bad == 0 and st[0] > 1000
    bad, checked = tw.arrmask_check(text, offs)
    assert bad == 0 and checked > 100000
    toks, toffs = tw.encode_batch(text, offs)
    assert np.array_equal(toffs, cl100k_golden["enc_offsets"]) and np.array_equal(toks, cl100k_golden["enc"])


def test_twin_fuzz_vs_restatement():
    tw, O = H.twin_cl100k(), H.port_tokenizer_cl100k()
    rng = random.Random(78)
    for it in range(12):
        docs = [e.encode("utf-8") for e in EDGE] if it == 0 else []
        for _ in range(rng.randint(1, 150)):
            r = rng.random()
            if r < 0.1:
                docs.append((rng.choice(["a", " ", "=", "1", "\n", "A", "x'S", "'ll", "é", "/\n"]) * rng.randint(50, 6000)).encode())
            else:
                docs.append("".join(H.fuzz_string(rng) for _ in range(rng.randint(1, 30))).encode("utf-8"))
        text, offs = H.pack_docs(docs)
        toks, toffs = tw.encode_batch(text, offs)
        etoks, eoffs = O.encode_batch(text, offs)
        assert np.array_equal(toffs, eoffs) and np.array_equal(toks, etoks)
        bad, _ = tw.sync_violations(text, offs)
        assert bad == 0


@pytest.mark.gpu
def test_gpu_cl100k_style_parity(golden, cl100k_golden):
    import td_corpus
    from tokendagger_amd import capi
    _, mr, special = H.llama4()
    tok = capi.HipTokenizer(H.CL100K_PAT, mr, special, device=0)
    tok2 = capi.HipTokenizer(vocab_io.CL100K_PAT_STR_POSSESSIVE, mr, special, device=0)
    text, offs = golden["text"], golden["offsets"]
    for t in (tok, tok2):
        toks, toffs = t.encode_batch(text, offs)
        assert np.array_equal(toffs, cl100k_golden["enc_offsets"]) and np.array_equal(toks, cl100k_golden["enc"])
    O = H.port_tokenizer_cl100k()
    rng = random.Random(22)
    for it in range(6):
        docs = ["".join(H.fuzz_string(rng) for _ in range(rng.randint(1, 40))).encode("utf-8") for _ in range(300)]
        docs += [(rng.choice(["7", "it's ", "'re", "é ", "a/\n"]) * rng.randint(100, 5000)).encode() for _ in range(4)]
        docs += [e.encode("utf-8") for e in EDGE]
        t, o = H.pack_docs(docs)
        toks, toffs = tok.encode_batch(t, o)
        etoks, eoffs = O.encode_batch(t, o)
        assert np.array_equal(toffs, eoffs) and np.array_equal(toks, etoks)
    for gen in (td_corpus.mixed, td_corpus.code, td_corpus.english):
        x, o = gen(2 << 20, seed=7)
        toks, toffs = tok.encode_batch(x, o)
        etoks, eoffs = O.encode_batch(x.tobytes(), o)
        assert np.array_equal(toffs, eoffs) and np.array_equal(toks, etoks)


@pytest.mark.gpu
def test_gpu_current_tiktoken_spelling_is_its_own_variant():
    """`\\s++$` ahead of `\\s*[\\r\\n]` (cl100k_base in current tiktoken releases): trailing whitespace with CR/LF is one
    piece.  GPU ids == the restatement's variant (which the CPU suite pins against PCRE2 running that very pattern)."""
    from tokendagger_amd import capi
    from oracle import port
    _, mr, special = H.llama4()
    tok = capi.HipTokenizer(vocab_io.CL100K_PAT_STR_CURRENT, mr, special, device=0)
    O = port.OracleTokenizer(mr, port.VARIANT_CL100K_EOS)
    rng = random.Random(23)
    tails = ["\r\t", "\n ", " \n\t ", "\r\n\r\n  ", "\t", "  ", "\n", " \r", "x\n\t", " \n "]
    for it in range(4):
        docs = ["".join(H.fuzz_string(rng) for _ in range(rng.randint(1, 40))) + rng.choice(tails) * rng.randint(0, 3) for _ in range(400)]
        docs += [("a" * rng.randint(1, 9000)) + rng.choice(tails) * rng.randint(1, 40) for _ in range(6)]  # tails across tile seams
        docs += [e + t for e in EDGE[:40] for t in tails[:3]]
        t, o = H.pack_docs([d.encode("utf-8") for d in docs])
        toks, toffs = tok.encode_batch(t, o)
        etoks, eoffs = O.encode_batch(t, o)
        assert np.array_equal(toffs, eoffs) and np.array_equal(toks, etoks)
    tok.close()
