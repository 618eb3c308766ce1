"""Round 5: a missed piece is merged once per call however often its bytes occur (td_collect_misses looks the pieces of the tiles
with many missed pieces up in a table of the call's distinct ones, td_copy_dups copies the ids; TD_OPT_DEDUPE).  Equal bytes have equal
ids — bpe_merge reads nothing but the piece (/root/reference/src/tiktoken/tiktoken.cpp:298-368) — so the ids must not change: checked
against the compiled reference and against the same call with the table switched off, on text that repeats its missed pieces a lot
(mixed-script text, the reference's code file set, emoji), on text that never does (random words), with a table of TWO seats (nearly
every piece finds both taken and is merged itself), with 64 seats (collisions: different pieces on one seat, told apart by their bytes)
and with lists so short that the repeats' originals are merged by the scan behind the rows."""
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


def _tok(**env):
    pat, mr, special = H.llama4()
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        t = capi.HipTokenizer(pat, mr, special, device=0)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    t.set_option(capi.TD_OPT_SMALL_PATH, 0)
    return t


def _inputs():
    rng = random.Random(21)
    out = {}
    x, o = td_corpus.mixed(3 << 20, seed=5)
    out["mixed-script 3 MiB"] = (x.tobytes(), o)
    x, o = td_corpus.code_files(3 << 20)
    out["code file set"] = (x.tobytes(), o)
    emoji = "".join(rng.choice("😀🎉👨‍💻🇩🇪✨🔥 aé中") for _ in range(60000)).encode()
    out["emoji"] = (emoji, [0, len(emoji)])
    # words that are no tokens and (nearly) never repeat; a few lengths around the 16-byte steps of the comparison
    words = []
    for _ in range(120000):
        n = rng.choice((2, 3, 5, 8, 9, 15, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64))
        words.append("".join(rng.choice("qxzjkvwy") for _ in range(n)))
    out["random words"] = (" ".join(words).encode(), [0, len(" ".join(words))])
    # the same few words over and over, cut into documents at odd places (a piece at the end of a document is a different piece)
    few = ["zzxqj", "Ünïcödé", "qqqqqqqqqqqqqqqqq", "日本語のテキスト", "xkcdxkcdxkcdxkcdxkcdxkcdxkcdxkcdxkcd"]
    rep = " ".join(rng.choice(few) for _ in range(150000)).encode()
    cuts = sorted({0, len(rep), *[c for c in (rng.randrange(len(rep)) for _ in range(500)) if (rep[c] & 0xC0) != 0x80]})
    out["five words"] = (rep, cuts)
    return out


def _reference(text: bytes, offs):
    R = H.ref_tokenizer()
    _, et, eo = R.encode_batch(np.frombuffer(text, dtype=np.uint8), np.asarray(offs, dtype=np.int64), n_threads=os.cpu_count() or 1, want_tokens=True)
    return et, eo


def test_dedupe_changes_no_id():
    assert ref.available(), "oracle/_ref/libtdref.so is missing: build it where /root/reference exists (oracle/build_ref.sh)"
    inputs = _inputs()
    want = {k: _reference(*v) for k, v in inputs.items()}
    toks = {"default": _tok(), "two seats": _tok(TD_DD_ENTRIES=2), "64 seats": _tok(TD_DD_ENTRIES=64),
            "short lists": _tok(TD_COLL_SHRINK=100000), "short lists, 64 seats": _tok(TD_COLL_SHRINK=100000, TD_DD_ENTRIES=64)}
    try:
        for name, t in toks.items():
            for fused in (1, 0):
                t.set_option(capi.TD_OPT_FUSED, fused)
                for dd in (1, 0):
                    t.set_option(capi.TD_OPT_DEDUPE, dd)
                    for k, (text, offs) in inputs.items():
                        gt, go = t.encode_batch(text, np.asarray(offs, dtype=np.int64))
                        et, eo = want[k]
                        assert np.array_equal(go, eo), f"{name}, fused={fused}, dedupe={dd}, {k}: document offsets differ from the reference"
                        bad = np.flatnonzero(gt != et) if len(gt) == len(et) else np.asarray([min(len(gt), len(et))])
                        assert bad.size == 0, f"{name}, fused={fused}, dedupe={dd}, {k}: ids differ from the reference, first at token {bad[:1]}"
                        # ... and the table does what it is there for (a switch that silently did nothing would pass everything above)
                        rep, listed = t.info(capi.TD_INFO_REPEATS), t.info(capi.TD_INFO_LISTED_PIECES)
                        if not dd:
                            assert rep == 0, f"{name}, {k}: repeats with the table off"
                        elif name == "default" and k in ("five words", "code file set", "mixed-script 3 MiB"):
                            assert rep > 4 * listed, f"{name}, fused={fused}, {k}: {rep} repeats, {listed} pieces merged themselves"
                        elif name == "default" and k == "random words":
                            assert rep < listed // 4, f"{name}, {k}: {rep} repeats among random words ({listed} merged)"
    finally:
        for t in toks.values():
            t.close()


def test_dedupe_in_ordinary_mode_and_with_specials():
    """encode_ordinary (mode 1) and the device-side special search run the same kernels: same ids with the table on and off."""
    pat, mr, special = H.llama4()
    t = _tok()
    try:
        x, o = td_corpus.mixed(2 << 20, seed=9)
        a = t.encode_batch(x.tobytes(), o, mode=1)
        t.set_option(capi.TD_OPT_DEDUPE, 0)
        b = t.encode_batch(x.tobytes(), o, mode=1)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        x, o = td_corpus.chat(2 << 20, seed=3)
        ids = sorted(special.values())
        b = t.encode_batch_with_special(x.tobytes(), o, ids)
        t.set_option(capi.TD_OPT_DEDUPE, 1)
        a = t.encode_batch_with_special(x.tobytes(), o, ids)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    finally:
        t.close()
