"""The fused tile loop (td_split_tiles<.., true>: pre-tokenizer + whole-piece lookup + in-place merge of the few pieces
that are no token, one pass over the text) against the compiled reference AND against the two-kernel form it replaces
(TD_OPT_FUSED = 0), on the inputs that take each of its branches:

  plain halves (every piece a token)                      synthetic English
  halves with 1..16 missed pieces (merged in place)       English with rare words sprinkled in
  halves with more (handed to td_merge_pieces)            mixed-script text, emoji, random bytes
  pieces above 64 bytes (TOK_LONGREF)                     long runs, CJK sentences
  deferred token tiles: more than 2048 pieces in 4 KiB    "a.b.c." / "1 2 3 "
                        pieces longer than the window      runs of 300 .. 100 000 bytes, across tile boundaries
                        a tile that starts inside one
  tile geometry                                           sizes around multiples of 4096 / 8192, documents cut everywhere

Reference behaviour: CoreBPE::encode, /root/reference/src/tiktoken/tiktoken.cpp:169-234 (through oracle/_ref, the checker).
"""
from __future__ import annotations

import os
import random

import numpy as np
import pytest

import helpers as H
import td_corpus
from oracle import ref
from tokendagger_amd import capi

pytestmark = pytest.mark.gpu

TD_OPT_FUSED = capi.TD_OPT_FUSED


@pytest.fixture(scope="module")
def tok():
    pat, mr, special = H.llama4()
    t = capi.HipTokenizer(pat, mr, special, device=0)
    t.set_option(capi.TD_OPT_SMALL_PATH, 0)  # (the one-launch path for tiny inputs is not what is tested here)
    yield t
    t.close()


def _both(tok, text: bytes, offs):
    """-> (fused loop with direct placement, TD_OPT_DIRECT = 1), (two-kernel form); the fused loop with every tile staged (the
    default) is run in between, with the pair of pack kernels (the default) and with the single one, and must agree with the first."""
    offs = np.asarray(offs, dtype=np.int64)
    tok.set_option(TD_OPT_FUSED, 1)
    tok.set_option(capi.TD_OPT_DIRECT, 1)
    ft, fo = tok.encode_batch(text, offs)
    tok.set_option(capi.TD_OPT_DIRECT, 0)
    st, so = tok.encode_batch(text, offs)
    tok.set_option(capi.TD_OPT_PACK_SPLIT, 0)  # one pack kernel for all tiles (rounds 2-4) instead of td_pack_plain + td_pack_rest
    pt, po = tok.encode_batch(text, offs)
    tok.set_option(capi.TD_OPT_PACK_SPLIT, 1)
    assert np.array_equal(po, so) and np.array_equal(pt, st), "the pair of pack kernels and the single one differ"
    tok.set_option(TD_OPT_FUSED, 0)
    ut, uo = tok.encode_batch(text, offs)
    tok.set_option(TD_OPT_FUSED, 1)
    assert np.array_equal(fo, so), "document offsets differ between direct placement and the staged form of the fused loop"
    assert np.array_equal(ft, st), "ids differ between direct placement and the staged form of the fused loop"
    return (ft, fo), (ut, uo)


def _check(tok, text: bytes, offs, what: str):
    (ft, fo), (ut, uo) = _both(tok, text, offs)
    assert np.array_equal(fo, uo), f"{what}: document offsets differ between the fused and the two-kernel form"
    assert np.array_equal(ft, ut), f"{what}: ids differ between the fused and the two-kernel form"
    assert ref.available(), "oracle/_ref/libtdref.so is missing: build it where /root/reference exists (oracle/build_ref.sh)"
    R = H.ref_tokenizer()
    _, et, eo = R.encode_batch(np.frombuffer(text, dtype=np.uint8), np.asarray(offs, dtype=np.int64), n_threads=os.cpu_count() or 1,
                               want_tokens=True)
    assert np.array_equal(fo, eo), f"{what}: document offsets differ from the reference"
    bad = np.flatnonzero(ft != et)
    assert bad.size == 0, f"{what}: ids differ from the reference, first at token {bad[:1]}"
    return ft, fo


def _rare_words(n: int, seed: int, every: int) -> bytes:
    """English text with a word that is no token about every `every` bytes: halves with a handful of missed pieces."""
    rng = random.Random(seed)
    x, _ = td_corpus.english(n, seed=seed)
    b = bytearray(x.tobytes())
    out = bytearray()
    pos = 0
    while pos < len(b):
        step = rng.randrange(every // 2, every * 3 // 2)
        out += b[pos:pos + step]
        pos += step
        k = rng.randrange(3)
        if k == 0:
            out += (" " + "".join(rng.choice("qxzjkvw") for _ in range(rng.randrange(5, 40)))).encode()
        elif k == 1:
            out += (" " + rng.choice(["zxqvbn", "Xylophonyx", "qwrtzp", "ĉapelo", "żółć", "naïveté", "ǆǆǆ"]) + " ").encode()
        else:
            out += (" x" + "".join(rng.choice("0123456789abcdef") for _ in range(rng.randrange(8, 60)))).encode()
    return bytes(out)


def test_plain_text_and_tile_geometry(tok):
    x, o = td_corpus.english(3 << 20, seed=11)
    _check(tok, x.tobytes(), o, "english 3 MiB")
    base = x.tobytes()
    for n in (1, 15, 16, 17, 4095, 4096, 4097, 8191, 8192, 8193, 8192 + 127, 8192 + 129, 8192 + 191, 8192 + 193, 16384, 16385, 3 * 8192 - 1,
              65536 + 5):
        _check(tok, base[:n], [0, n], f"english, {n} bytes, one document")
    # documents cut everywhere (mid-word): every token tile sees document starts at odd places
    rng = random.Random(5)
    cuts = sorted(set(rng.randrange(0, 300000) for _ in range(4000)) | {0, 300000})
    _check(tok, base[:300000], cuts, "english, 4000 random document cuts")
    cuts = list(range(0, 40001))  # one-byte documents, including across tile boundaries
    _check(tok, base[:40000], cuts, "english, one-byte documents")
    _check(tok, base[:20000], [0, 0, 0, 5, 5, 8192, 8192, 8192 + 128, 20000, 20000], "empty documents")


def test_halves_with_a_few_missed_pieces_are_merged_in_place(tok):
    for every, seed in ((300, 1), (1200, 2), (4000, 3), (150, 4)):
        t = _rare_words(1 << 20, seed, every)
        _check(tok, t, [0, len(t)], f"rare words every ~{every} bytes")
        offs = sorted(set([0, len(t)] + [random.Random(seed).randrange(len(t)) for _ in range(500)]))
        _check(tok, t, offs, f"rare words every ~{every} bytes, 500 documents")


def test_heavy_halves_go_to_the_merge_kernel(tok, golden):
    x, o = td_corpus.mixed(2 << 20, seed=7)
    _check(tok, x.tobytes(), o, "mixed-script 2 MiB")
    x, o = td_corpus.code(2 << 20, seed=7)
    _check(tok, x.tobytes(), o, "synthetic code 2 MiB")
    rng = random.Random(9)
    emoji = "".join(rng.choice("😀🎉👨‍💻🇩🇪✨🔥 aé中") for _ in range(40000)).encode()
    _check(tok, emoji, [0, len(emoji)], "emoji")
    _check(tok, golden["text"].tobytes(), golden["offsets"], "golden documents")


def test_miss_lists_that_are_full_leave_their_tiles_to_the_scan_behind_the_rows(tok):
    """td_collect_misses puts the missed pieces of the flagged tiles on lists sized for a few times the density of real text; a
    class of a tile that finds no room is merged by the scan at the end of td_merge_pieces.  TD_COLL_SHRINK (read by td_create)
    makes the lists 1/k of their size: with k = 40 a part of the tiles overflow, with k = 10^6 nearly all of them (a list then
    holds 65 records), and the ids must not change."""
    pat, mr, special = H.llama4()
    x, o = td_corpus.mixed(3 << 20, seed=11)
    rng = random.Random(5)
    emoji = "".join(rng.choice("😀🎉👨‍💻🇩🇪✨🔥 aé中") for _ in range(30000)).encode()
    want = tok.encode_batch(x.tobytes(), o)
    want_e = tok.encode_batch(emoji, np.asarray([0, len(emoji)], dtype=np.int64))
    assert ref.available(), "oracle/_ref/libtdref.so is missing: build it where /root/reference exists (oracle/build_ref.sh)"
    _, et, eo = H.ref_tokenizer().encode_batch(x, np.asarray(o, dtype=np.int64), n_threads=os.cpu_count() or 1, want_tokens=True)
    assert np.array_equal(want[0], et) and np.array_equal(want[1], eo)
    for k in (40, 1000000):
        os.environ["TD_COLL_SHRINK"] = str(k)
        try:
            t2 = capi.HipTokenizer(pat, mr, special, device=0)
        finally:
            del os.environ["TD_COLL_SHRINK"]
        try:
            t2.set_option(capi.TD_OPT_SMALL_PATH, 0)
            for fused in (1, 0):
                t2.set_option(TD_OPT_FUSED, fused)
                got = t2.encode_batch(x.tobytes(), o)
                assert np.array_equal(got[1], want[1]) and np.array_equal(got[0], want[0]), f"lists 1/{k}, fused={fused}: mixed-script text"
                got = t2.encode_batch(emoji, np.asarray([0, len(emoji)], dtype=np.int64))
                assert np.array_equal(got[1], want_e[1]) and np.array_equal(got[0], want_e[0]), f"lists 1/{k}, fused={fused}: emoji"
        finally:
            t2.close()


def test_long_pieces_and_deferred_tiles(tok):
    rng = random.Random(3)
    filler, _ = td_corpus.english(1 << 18, seed=2)
    filler = filler.tobytes()
    cases = {
        "dense punctuation (> 2048 pieces per half)": b"a.b.c.d.e.f.g.h." * 4000,
        "digits and blanks": b"1 2 3 4 5 6 7 8 9 0 " * 3000,
        "every byte a piece": b".,;:!?()[]{}" * 5000,
        "run of 300": filler[:5000] + b"a" * 300 + filler[:5000],
        "runs around the halo sizes": b"".join(filler[i * 700:(i + 1) * 700] + bytes([97 + i % 26]) * L for i, L in
                                              enumerate([63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 256, 257, 1023, 1024, 1025])),
        "blanks 5000": filler[:9000] + b" " * 5000 + filler[:3000],
        "piece over several tiles": filler[:8000] + b"z" * 30000 + filler[:100] + b"=" * 20000 + b"\n" * 9000 + filler[:8000],
        "100 KB of one letter": b"q" * 100000,
        "cjk sentences": ("".join(rng.choice("的一是不了人我在有他这为之大来以个中上们") for _ in range(60000))).encode(),
        "cjk with punctuation": "".join("".join(rng.choice("的一是不了人我在有他这为之大来以个中上们") for _ in range(rng.randrange(1, 90))) + rng.choice("，。！？ \n")
                                        for _ in range(2500)).encode(),
    }
    for name, t in cases.items():
        _check(tok, t, [0, len(t)], name)
        # the same text at every alignment of the interesting stretch relative to the tile grid
    t = filler[:8192 - 200]
    for shift in (0, 1, 63, 64, 100, 127, 128, 129, 190, 200, 250):
        u = t + b" " * shift + b"x" * 400 + filler[:9000]
        _check(tok, u, [0, len(u)], f"run of 400 at tile offset -200+{shift}")


def test_random_text_and_invalid_utf8(tok):
    rng = random.Random(21)
    # valid UTF-8 from all over the code space (the reference runs PCRE2 with NO_UTF_CHECK: invalid UTF-8 is undefined
    # behaviour there — it crashes on random bytes — so only valid text is compared with it)
    uni = "".join(H.random_unicode_string(rng, 40) for _ in range(8000)).encode("utf-8")
    cuts = sorted(set([0, len(uni)]))
    _check(tok, uni, cuts, "random code points")
    ascii_junk = bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789 \n\t.,'") for _ in range(300000))
    _check(tok, ascii_junk, [0, len(ascii_junk)], "random ASCII")
    # random bytes: the fused and the two-kernel form must still agree with each other
    junk = bytes(rng.randrange(256) for _ in range(200000))
    offs = sorted(set([0, len(junk)] + [rng.randrange(len(junk)) for _ in range(300)]))
    (ft, fo), (ut, uo) = _both(tok, junk, offs)
    assert np.array_equal(fo, uo) and np.array_equal(ft, ut), "random bytes: fused and two-kernel ids differ"


def test_encode_ordinary_mode_takes_the_same_path(tok):
    x, o = td_corpus.english(1 << 19, seed=4)
    tok.set_option(TD_OPT_FUSED, 1)
    a = tok.encode_batch(x.tobytes(), o, mode=capi.TD_MODE_ORDINARY)
    tok.set_option(TD_OPT_FUSED, 0)
    b = tok.encode_batch(x.tobytes(), o, mode=capi.TD_MODE_ORDINARY)
    tok.set_option(TD_OPT_FUSED, 1)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_repeated_device_calls_replay_a_graph_with_the_same_results(tok):
    """td_encode_device: the second identical call in a row captures the step as a hipGraph, later ones replay it
    (TD_OPT_GRAPH).  Results must not depend on which way a call was launched, and a change of any argument must not
    replay the old graph."""
    import torch
    x, o = td_corpus.mixed(3 << 20, seed=5)
    y, p = td_corpus.english(2 << 20, seed=6)
    assert ref.available(), "oracle/_ref/libtdref.so is missing: build it where /root/reference exists (oracle/build_ref.sh)"
    R = H.ref_tokenizer()
    s = torch.cuda.current_stream().cuda_stream

    def run(text, offs, times):
        n, nd = len(text), len(offs) - 1
        dt, do = torch.from_numpy(text).cuda(), torch.from_numpy(offs).cuda()
        dk = torch.empty(n + 1024, dtype=torch.int32, device="cuda")
        dto = torch.empty(nd + 1, dtype=torch.int64, device="cuda")
        outs = []
        for _ in range(times):
            dk.zero_()
            dto.zero_()
            tok.encode_device(dt.data_ptr(), n, do.data_ptr(), nd, dk.data_ptr(), n + 1024, dto.data_ptr(), s)
            tok.device_status(s)
            toff = dto.cpu().numpy()
            outs.append((dk[:int(toff[-1])].cpu().numpy(), toff))
        return outs

    for g in (1, 0):
        tok.set_option(capi.TD_OPT_GRAPH, g)
        for text, offs in ((x, o), (y, p), (x, o)):
            outs = run(text, offs, 5)
            for t_, o_ in outs[1:]:
                assert np.array_equal(t_, outs[0][0]) and np.array_equal(o_, outs[0][1]), "a replayed step gave different ids"
            if R is not None:
                _, et, eo = R.encode_batch(text, offs, n_threads=os.cpu_count() or 1, want_tokens=True)
                assert np.array_equal(outs[-1][1], eo) and np.array_equal(outs[-1][0], et)
    tok.set_option(capi.TD_OPT_GRAPH, 0)  # (the default)


def test_direct_placement_takes_the_tiles_it_can_and_stages_the_rest(tok):
    """Round 4, opt-in (TD_OPT_DIRECT = 1; measured slower than the staged form on this chip): the fused
    loop writes a tile's ids straight to the output when the tile's base is known in time (decoupled look-back).  Plain
    English: nearly every tile; English with a rare word per tile: still (the few missed pieces are merged inside the loop);
    a piece above 64 bytes or a tile full of missed pieces breaks the chain: the tiles in front of it are placed directly,
    the ones behind it are staged — same ids either way (every _check compares the two)."""
    x, o = td_corpus.english(64 << 20, seed=91)
    _check(tok, x.tobytes(), o, "english 64 MiB")
    n8 = (len(x) + 8191) // 8192
    assert tok.info(capi.TD_INFO_DIRECT_TILES) == 0  # (_both ends on the staged forms: the counter is the last call's)
    tok.set_option(capi.TD_OPT_DIRECT, 1)
    tok.encode_batch(x.tobytes(), o)
    # (a workgroup's last tiles are placed without the delay that lets their predecessors publish: of the five or six
    # tiles a workgroup has here some are staged; 1024 MiB: 98 % direct)
    assert tok.info(capi.TD_INFO_DIRECT_TILES) >= n8 // 4, (tok.info(capi.TD_INFO_DIRECT_TILES), n8)
    rare = _rare_words(6 << 20, 92, 3000)
    _check(tok, rare, [0, len(rare)], "rare words, one document")
    tok.set_option(capi.TD_OPT_DIRECT, 1)
    tok.encode_batch(rare, np.asarray([0, len(rare)], dtype=np.int64))
    assert tok.info(capi.TD_INFO_DIRECT_TILES) > 0
    # the chain breaks in the middle: a run of 200 bytes (a piece above 64 bytes) at 3 MiB
    b = bytearray(x[:6 << 20].tobytes())
    b[3 << 20:(3 << 20) + 200] = b"=" * 200
    oo = o[o < (6 << 20)]
    oo = np.concatenate([oo, [6 << 20]]) if oo[-1] != (6 << 20) else oo
    _check(tok, bytes(b), oo, "chain broken at 3 MiB")
    tok.set_option(capi.TD_OPT_DIRECT, 1)
    tok.encode_batch(bytes(b), oo)
    d = tok.info(capi.TD_INFO_DIRECT_TILES)
    assert 0 < d <= (3 << 20) // 8192 + 1, d
    # capacity: nothing is written past the end of a too small output, the needed size comes back
    with pytest.raises(capi.TokenDaggerHipError) as e:
        tok.encode_batch(x[:1 << 20].tobytes(), o[o <= (1 << 20)], capacity=1000)
    assert e.value.code == capi.TD_E_CAPACITY
    tok.set_option(capi.TD_OPT_DIRECT, 0)  # (the default)
