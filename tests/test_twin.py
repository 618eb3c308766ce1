"""Host logic of the product, run on the CPU through the twin harness (tests/twin/td_twin.cpp): the same
header code the kernels compile (classification, sync points, scanner, tile/lane speculation) and the host
table builder, checked against golden vectors and the oracle."""
import random

import numpy as np
import pytest

import helpers as H
from oracle import port


def _docs(golden, step=1):
    text, offs = golden["text"].tobytes(), golden["offsets"]
    for d in range(0, len(offs) - 1, step):
        yield d, text[offs[d]:offs[d + 1]]


def test_table_sizes():
    tw = H.twin_llama4()
    assert tw.info(1) == 439802, "pair table entries for Llama-4 (SURVEY 7-A ii)"
    assert tw.info(3) == 201133
    # specials are inserted into mergeable_ranks by the tests -> unreachable by merging -> not closed
    assert tw.info(2) == 0
    pat, mr, special = H.llama4()
    plain = {k: v for k, v in mr.items() if v < 200000}
    assert H.Twin(pat, plain, special).info(2) == 1, "plain Llama-4 vocab: every token is merge-reachable"


def test_unsupported_pattern_is_rejected():
    with pytest.raises(RuntimeError):
        H.Twin(r"[a-zA-Z]+|\s+|[0-9]+|[^\w\s]", {b"a": 0})


def test_serial_scanner_matches_golden_pieces(golden):
    tw = H.twin_llama4()
    pe, po = golden["piece_ends"], golden["piece_offsets"]
    for d, doc in _docs(golden):
        if not doc:
            continue
        ends = pe[po[d]:po[d + 1]]
        starts = np.concatenate([[0], ends[:-1]])
        assert np.array_equal(tw.split_serial(doc), starts), golden["names"][d]


def test_tiled_speculative_scan_on_whole_golden_batch(golden):
    # all golden docs as ONE batch: exercises tile boundaries, document flags, long pieces, slow path
    tw = H.twin_llama4()
    text, offs = golden["text"].tobytes(), golden["offsets"]
    pe, po = golden["piece_ends"], golden["piece_offsets"]
    exp = []
    for d in range(len(offs) - 1):
        ends = pe[po[d]:po[d + 1]]
        if len(ends):
            exp.append(np.concatenate([[0], ends[:-1]]) + offs[d])
    exp = np.concatenate(exp)
    got, stats = tw.split_tiled(text, offs)
    assert np.array_equal(got, exp)
    assert stats[0] > 0, "golden set contains pieces that leave their tile window"
    assert stats[1] > 0, "... and tiles that lie inside a piece longer than the left halo"
    assert 0 < stats[4] < stats[3] // 2, "the whole-word rules resolve most synchronisation points"


def test_sync_points_are_always_piece_starts(golden):
    tw = H.twin_llama4()
    text, offs = golden["text"].tobytes(), golden["offsets"]
    bad, n = tw.sync_violations(text, offs)
    assert bad == 0 and n > 100000
    rng = random.Random(5)
    for i in range(4000):
        s = (H.fuzz_string(rng, 30) if i % 2 else H.random_unicode_string(rng, 60)).encode("utf-8")
        bad, _ = tw.sync_violations(s)
        assert bad == 0, repr(s)


def test_whole_word_rules_only_resolve_single_pieces(golden):
    """split_unresolved_heads (carry arithmetic on 64-byte mask windows): a head it calls resolved is one piece that ends
    at the next synchronisation point; on running text it resolves most heads."""
    text, offs = golden["text"].tobytes(), golden["offsets"]
    for tw in (H.twin_llama4(), H.twin_tekken()):
        bad, st = tw.word_rules_check(text, offs)
        assert bad == 0 and st[0] > 100000 and st[1] < st[0] // 2
        rng = random.Random(11)
        for i in range(6000):
            s = (H.fuzz_string(rng, 90) if i % 2 else H.random_unicode_string(rng, 120)).encode("utf-8")
            if i % 7 == 0:
                s = bytes(rng.randrange(256) for _ in range(rng.randrange(1, 200)))  # not UTF-8
            if s:
                bad, _ = tw.word_rules_check(s)
                assert bad == 0, repr(s)


def test_twin_encode_matches_golden(golden):
    tw = H.twin_llama4()
    text, offs = golden["text"].tobytes(), golden["offsets"]
    toks, toffs = tw.encode_batch(text, offs, mode=0)
    assert np.array_equal(toffs, golden["enc_offsets"])
    assert np.array_equal(toks, golden["enc"])
    toks1, toffs1 = tw.encode_batch(text, offs, mode=1)
    assert np.array_equal(toks1, golden["enc"]) and np.array_equal(toffs1, toffs)


def test_twin_vs_oracle_fuzz_batches():
    tw, O = H.twin_llama4(), H.port_tokenizer()
    rng = random.Random(99)
    for _ in range(20):
        docs = []
        for _ in range(rng.randint(1, 200)):
            r = rng.random()
            if r < 0.05:
                docs.append(b"")
            elif r < 0.1:
                docs.append((rng.choice(["a", " ", "=", "1", "\n", "A", "xY"]) * rng.randint(50, 6000)).encode())
            else:
                docs.append("".join(H.fuzz_string(rng) for _ in range(rng.randint(1, 30))).encode("utf-8"))
        text, offs = H.pack_docs(docs)
        toks, toffs = tw.encode_batch(text, offs)
        etoks, eoffs = O.encode_batch(text, offs)
        assert np.array_equal(toffs, eoffs)
        assert np.array_equal(toks, etoks)


def test_incomplete_vocab_semantics():
    pat, _, _ = H.llama4()
    vocab = {b"a": 0, b"b": 1, b"ca": 2, b"ab": 3}
    tw = H.Twin(pat, vocab)
    O = port.OracleTokenizer(vocab)
    assert tw.encode_batch(b"ab")[0].tolist() == [3]
    # a non-token byte may still merge into a token: the reference keys get_rank by bytes (tiktoken.cpp:282-296)
    assert tw.encode_batch(b"ca")[0].tolist() == [2] and O.encode(b"ca").tolist() == [2]
    assert tw.encode_batch(b"cab")[0].tolist() == [2, 1] and O.encode(b"cab").tolist() == [2, 1]
    for bad in (b"c", b"abc", b"bc"):
        with pytest.raises(RuntimeError):
            tw.encode_batch(bad)
        with pytest.raises(port.OracleError):
            O.encode(bad)


def test_pair_table_probed_in_the_devices_order():
    """td_merge_pieces probes a pair's first seat and its second one only where the first neither holds the pair nor is marked
    final (PAIR_FINAL, set by build_tables for the slots nobody was pushed out of); a part is set up from ONE load of
    Tables::byte_pair_id.  Both against the plain tables: every pair of the vocabulary, every one of them with its halves
    swapped, random id pairs."""
    for tw in (H.twin_llama4(), H.twin_gpt2()):
        bad, st = tw.pair_probe_check(300000, seed=3)
        assert bad == 0, st
        assert st[0] > 1000 and st[4] == 0
        assert st[1] < 0.4 * st[0], f"more than 40 % of the pairs sit in their second seat: {st}"
        assert st[3] < 0.25 * st[2], f"the first seat is final for fewer than three in four of the pairs that are none: {st}"
