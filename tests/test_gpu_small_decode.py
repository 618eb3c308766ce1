"""decode_bytes on at most 1024 ids is ONE launch over pinned host buffers (td_small_decode; reference: CoreBPE::decode_bytes,
/root/reference/src/tiktoken/tiktoken.cpp:236-255).  Same bytes and the same errors as the general three-launch path."""
import random

import numpy as np
import pytest

import helpers as H
from tokendagger_amd import capi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tok():
    pat, mr, special = H.llama4()
    t = capi.HipTokenizer(pat, mr, special, device=0)
    yield t
    t.close()


def test_small_decode_equals_the_general_path_and_the_vocabulary(tok):
    pat, mr, special = H.llama4()
    by_id = {r: b for b, r in mr.items()}
    by_id.update({i: s.encode() for s, i in special.items()})
    rng = random.Random(4)
    all_ids = sorted(by_id)
    longest = sorted(all_ids, key=lambda i: -len(by_id[i]))[:40]
    cases = [[all_ids[0]], [rng.choice(all_ids) for _ in range(7)], [rng.choice(all_ids) for _ in range(1023)],
             [rng.choice(all_ids) for _ in range(1024)], [rng.choice(all_ids) for _ in range(1025)],
             [rng.choice(longest) for _ in range(1024)],          # more than 16 KiB of bytes: handed to the general path
             [rng.choice(longest) for _ in range(150)], list(special.values())[:50]]
    for ids in cases:
        want = b"".join(by_id[i] for i in ids)
        arr = np.asarray(ids, dtype=np.int32)
        tok.set_option(capi.TD_OPT_SMALL_PATH, 1)
        a = tok.decode_bytes(arr)
        tok.set_option(capi.TD_OPT_SMALL_PATH, 0)
        b = tok.decode_bytes(arr)
        tok.set_option(capi.TD_OPT_SMALL_PATH, 1)
        assert a == want and b == want, len(ids)


def test_small_decode_reports_the_bad_id(tok):
    _, mr, special = H.llama4()
    bad = max(max(mr.values()), max(special.values())) + 17
    for ids in ([5, 6, bad, 7], [bad], [1] * 1000 + [-3]):
        for small in (1, 0):
            tok.set_option(capi.TD_OPT_SMALL_PATH, small)
            with pytest.raises(capi.TokenDaggerHipError) as e:
                tok.decode_bytes(np.asarray(ids, dtype=np.int32))
            assert e.value.code == 8 and f"Invalid token for decoding: {ids[-1] if ids[-1] < 0 else bad}" in str(e.value), (ids[:4], small, str(e.value))
    tok.set_option(capi.TD_OPT_SMALL_PATH, 1)
    # the handle still works after an error
    assert tok.decode_bytes(np.asarray([mr[b"hello"]], dtype=np.int32)) == b"hello"
