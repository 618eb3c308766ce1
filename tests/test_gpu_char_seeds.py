"""Character seeds on the GPU (td_long_pieces' set-up, lp_seed_setup; td_common.h: character seeds): pieces of 65..255 bytes in scripts
whose characters are tokens — CJK, kana, hangul, Thai, Devanagari, Cyrillic, accented Latin — against the compiled reference
(CoreBPE::encode, /root/reference/src/tiktoken/tiktoken.cpp:169-234, byte-pair merge :298-368), with the pieces at every alignment, at
the very start and the very end of the text (the set-up reads a 20-byte window per lane: its edge path), as one batch and one document
per call, and with the seeds switched off (TD_CHAR_SEEDS=0) for the same ids."""
from __future__ import annotations

import os
import random

import numpy as np
import pytest

import helpers as H
from oracle import ref
from tokendagger_amd import capi

pytestmark = pytest.mark.gpu


def _runs(rng, n):
    ranges = [(0x4E00, 0x9FFF), (0x3040, 0x30FF), (0xAC00, 0xD7A3), (0x0E01, 0x0E5B), (0x0900, 0x097F), (0x0400, 0x04FF), (0x00C0, 0x024F), (0x0600, 0x06FF)]
    common = "的一是不了人我在有他这为之大来以个中上们到说国和地也子时道出而要于就下得可你年生こんにちはありがとう世界カタカナ한국어안녕하세요감사합니다"
    out = []
    for _ in range(n):
        kind = rng.random()
        lo, hi = rng.choice(ranges)
        target = rng.randint(60, 260)
        s = ""
        while len(s.encode("utf-8")) < target:
            s += rng.choice(common) if kind < 0.5 else chr(rng.randint(lo, hi)) if kind < 0.9 else rng.choice([chr(rng.randint(lo, hi)), rng.choice(common), "a", "é"])
        out.append(s)
    return out


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


def test_long_pieces_of_seedable_characters_equal_the_reference():
    assert ref.available(), "oracle/_ref/libtdref.so is missing: build it where /root/reference exists (oracle/build_ref.sh)"
    R = H.ref_tokenizer()
    rng = random.Random(17)
    on, off = _tok(), _tok(TD_CHAR_SEEDS=0)
    assert on.info(capi.TD_INFO_CHAR_SEEDS) > 5000 and off.info(capi.TD_INFO_CHAR_SEEDS) == 0, "the Llama-4 vocabulary has 5 675 seedable characters"
    try:
        for rep in range(6):
            runs = _runs(rng, 1500)
            # as documents of their own (a run is one piece: letters only), glued with blanks / newlines / nothing at odd alignments,
            # and with a run as the very first and the very last bytes of the text
            docs = []
            for i, r in enumerate(runs):
                sep = rng.choice(["", " ", "\n", "。", ", ", "x" * rng.randint(0, 5)])
                docs.append((r + sep).encode("utf-8") if i + 1 < len(runs) else r.encode("utf-8"))
            text, offs = H.pack_docs(docs)
            _, et, eo = R.encode_batch(np.frombuffer(text, dtype=np.uint8), offs, n_threads=os.cpu_count() or 1, want_tokens=True)
            for name, t in (("seeds", on), ("no seeds", off)):
                gt, go = t.encode_batch(text, offs)
                assert np.array_equal(go, eo), f"{name}: document offsets differ from the reference (batch {rep})"
                bad = np.flatnonzero(gt != et)
                assert bad.size == 0, f"{name}: ids differ from the reference, first at token {int(bad[0])} (batch {rep})"
            # the same bytes as ONE document (pieces now join across the old document ends)
            whole = np.asarray([0, len(text)], dtype=np.int64)
            _, et, eo = R.encode_batch(np.frombuffer(text, dtype=np.uint8), whole, n_threads=1, want_tokens=True)
            gt, go = on.encode_batch(text, whole)
            assert np.array_equal(go, eo) and np.array_equal(gt, et), f"one document (batch {rep})"
            # single calls: the run is the whole text (both edges of the window path)
            for r in runs[:40]:
                b = r.encode("utf-8")
                want = R.encode(b)
                assert np.array_equal(on.encode(b), want), f"single call differs from the reference: {r!r}"
    finally:
        on.close()
        off.close()
