"""td_encode_batch on host buffers of 4 KiB .. 4 MiB (round 6, encode_batch_mid in td_api.cpp): one pinned buffer in, the step's pack kernels
writing ids and offsets straight into pinned host memory, a sequence number instead of stream synchronisations.  Same ids, offsets, counts
and errors as the copies-and-synchronise path of rounds 1-5 (TD_MID_PATH=0) and as the compiled reference (CoreBPE::encode,
/root/reference/src/tiktoken/tiktoken.cpp:169-234), at the sizes around both ends of the range and on every kind of text."""
from __future__ import annotations

import os

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
        return capi.HipTokenizer(pat, mr, special, device=0)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


def test_mid_size_host_batches_equal_the_reference_and_the_old_path():
    assert ref.available(), "oracle/_ref/libtdref.so is missing: build it where /root/reference exists (oracle/build_ref.sh)"
    R = H.ref_tokenizer()
    new, old = _tok(), _tok(TD_MID_PATH=0)
    try:
        rare = ("q" * 300 + " " + "ab" * 900 + "\n" + "中文" * 500 + "\n" + "7" * 9000 + "\n").encode()
        for name, gen in (("english", td_corpus.english), ("mixed", td_corpus.mixed), ("code", td_corpus.code)):
            for size in (4097, 5000, 65536, 300_000, (1 << 20) + 17, (4 << 20) - 4096, (4 << 20) + 4096):
                x, o = gen(size, seed=size % 97)
                if size == 300_000:  # (far pieces, long and giant pieces inside a mid-size call)
                    x = np.concatenate([x[:100_000], np.frombuffer(rare, dtype=np.uint8), x[100_000:]])
                    o = np.unique(np.concatenate([o[o <= 100_000], [100_000, 100_000 + len(rare)], o[o > 100_000] + len(rare)])).astype(np.int64)
                _, et, eo = R.encode_batch(x, o, n_threads=os.cpu_count() or 1, want_tokens=True)
                for rep in range(2):
                    gt, go = new.encode_batch(x.tobytes(), o)
                    assert np.array_equal(go, eo), f"{name} {size}: document offsets differ from the reference"
                    assert np.array_equal(gt, et), f"{name} {size}: ids differ from the reference"
                ot, oo = old.encode_batch(x.tobytes(), o)
                assert np.array_equal(ot, gt) and np.array_equal(oo, go), f"{name} {size}: the two host paths differ"
                # as ONE document (what enc.encode(text) of a file is)
                one = np.asarray([0, len(x)], dtype=np.int64)
                _, et1, eo1 = R.encode_batch(x, one, n_threads=1, want_tokens=True)
                gt1, go1 = new.encode_batch(x.tobytes(), one)
                assert np.array_equal(go1, eo1) and np.array_equal(gt1, et1), f"{name} {size}: one document"
    finally:
        new.close()
        old.close()


def test_mid_size_errors_and_capacity():
    """A byte that is no token in a vocabulary without it raises the same error with the same position; a too small output reports the count."""
    from tokendagger_amd import capi as C
    toy = {bytes([b]): b for b in range(97, 123)}
    toy[b" "] = 26
    toy[b"ab"] = 27
    pat, _, _ = H.llama4()
    tok = C.HipTokenizer(pat, toy, {}, device=0)
    try:
        good = (b"ab ba abab " * 1000)
        ids, offs = tok.encode_batch(good, np.asarray([0, len(good)], dtype=np.int64))
        assert tok.decode_bytes(ids) == good
        bad = good[:7000] + b"Z" + good[7000:]
        with pytest.raises(Exception) as ei:
            tok.encode_batch(bad, np.asarray([0, len(bad)], dtype=np.int64))
        assert "not in the vocabulary" in str(ei.value), str(ei.value)
        # the handle is usable afterwards
        ids2, _ = tok.encode_batch(good, np.asarray([0, len(good)], dtype=np.int64))
        assert np.array_equal(ids, ids2)
    finally:
        tok.close()
