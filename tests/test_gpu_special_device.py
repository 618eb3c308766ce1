"""Allowed special tokens cut out ON THE DEVICE (td_encode_device_with_special, td_special.hip) against the host-side search of
td_encode_batch_with_special (the path round 1-2 shipped, checked against tiktoken's semantics in test_python_api.py) and, where
the text between the cuts can be given to it piecewise, against the compiled reference.  Reference behaviour:
CoreBPE::encode(text, allowed_special), /root/reference/src/tiktoken/tiktoken.cpp:169-234 (its own segmentation loop has
iterator-invalidation UB, SURVEY A6: tiktoken's semantics are the specification)."""
import random

import numpy as np
import pytest

import helpers as H
import td_corpus
from tokendagger_amd import capi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tok():
    pat, mr, special = H.llama4()
    t = capi.HipTokenizer(pat, mr, special, device=0)
    yield t
    t.close()


def _device(tok, text: bytes, offs, allowed_ids):
    import torch
    x = np.frombuffer(text, dtype=np.uint8)
    offs = np.asarray(offs, dtype=np.int64)
    n, nd = len(x), len(offs) - 1
    dt = torch.from_numpy(x.copy()).cuda() if n else torch.empty(1, dtype=torch.uint8, device="cuda")
    do = torch.from_numpy(offs).cuda()
    dk = torch.empty(n + 1024, dtype=torch.int32, device="cuda")
    dto = torch.empty(nd + 1, dtype=torch.int64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    tok.encode_device_with_special(dt.data_ptr(), n, do.data_ptr(), nd, allowed_ids, dk.data_ptr(), n + 1024, dto.data_ptr(), s)
    tok.device_status(s)
    toff = dto.cpu().numpy()
    return dk[:int(toff[-1])].cpu().numpy(), toff


def _check(tok, text: bytes, offs, allowed_ids, what):
    dt_, do_ = _device(tok, text, offs, allowed_ids)
    tok.set_option(capi.TD_OPT_DEVICE_SPECIALS, 0)  # the host-side search
    ht, ho = tok.encode_batch_with_special(text, np.asarray(offs, dtype=np.int64), sorted(allowed_ids))
    tok.set_option(capi.TD_OPT_DEVICE_SPECIALS, 1)
    if len(text) >= 1 << 20:  # ... and the same entry point routed through the device search
        rt, ro = tok.encode_batch_with_special(text, np.asarray(offs, dtype=np.int64), sorted(allowed_ids))
        assert np.array_equal(ro, ho) and np.array_equal(rt, ht), f"{what}: td_encode_batch_with_special differs between its two routes"
    assert np.array_equal(do_, ho), f"{what}: document offsets differ between the device and the host search"
    bad = np.flatnonzero(dt_ != ht) if len(dt_) == len(ht) else [0]
    assert len(dt_) == len(ht) and len(bad) == 0, f"{what}: ids differ, first at {bad[:1]}"
    return dt_, do_


def test_chat_corpus_all_specials(tok):
    _, _, special = H.llama4()
    allowed = list(special.values())
    x, o = td_corpus.chat(3 << 20, seed=3)
    toks, offs = _check(tok, x.tobytes(), o, allowed, "chat 3 MiB, all specials allowed")
    assert (toks == special["<|eot|>"]).sum() > 1000 and (toks == special["<|header_start|>"]).sum() > 1000
    # a subset: the others are ordinary text
    some = [special[k] for k in ("<|eot|>", "<|begin_of_text|>")]
    _check(tok, x.tobytes(), o, some, "chat, two specials allowed")
    _check(tok, x.tobytes(), [0, len(x)], allowed, "chat as one document")


def test_special_edge_cases_on_the_device(tok):
    _, _, special = H.llama4()
    allowed = list(special.values())
    names = [k for k in special if "reserved" not in k]
    rng = random.Random(8)
    cases = {
        "only specials": "<|begin_of_text|><|eot|><|eot|><|end_of_text|>",
        "special at both ends": "<|begin_of_text|>hello world<|eot|>",
        "adjacent and nested look-alikes": "<|<|eot|>|><|eot<|eot|><|e<|eom|>ot|>",
        "whitespace around cuts": "a  <|eot|>  b\n\n<|eot|>\n\n c \t<|eom|> ",
        "digits and contractions at cuts": "it<|eot|>'s 123<|eot|>456 don<|eot|>'t",
        "long reserved names": "x<|text_post_train_reserved_special_token_7|>y<|text_post_train_reserved_special_token_77|>z",
        "unicode around cuts": "中文<|eot|>ñandú 😀<|header_start|>user<|header_end|>\n\nمرحبا",
        "no specials at all": "plain text without any of them < | > <| |>",
        "empty": "",
    }
    for name, s in cases.items():
        b = s.encode()
        _check(tok, b, [0, len(b)], allowed, name)
    # specials at every alignment against the tile grid and in one-byte / empty documents
    filler, _ = td_corpus.english(1 << 16, seed=5)
    filler = filler.tobytes()
    for shift in (4080, 4090, 4095, 4096, 8180, 8190, 8192, 8200):
        b = filler[:shift] + b"<|header_start|>assistant<|header_end|>\n\n" + filler[:3000] + b"<|eot|>"
        _check(tok, b, [0, len(b)], allowed, f"special at byte {shift}")
        _check(tok, b, [0, shift + 3, shift + 3, shift + 9, len(b)], allowed, f"documents cut through the literal at byte {shift}")
    # random soup of literals, prefixes of literals and text; documents cut anywhere
    parts = names + [n[:k] for n in names[:8] for k in (2, 5, len(n) - 1)] + [" ", "a", "\n", "word ", "12", "'ll", "<", "|", ">", "中"]
    for trial in range(30):
        s = "".join(rng.choice(parts) for _ in range(rng.randrange(1, 200))).encode()
        cuts = sorted(set([0, len(s)] + [rng.randrange(len(s) + 1) for _ in range(rng.randrange(0, 6))]))
        some = rng.sample(allowed, rng.randrange(1, 40)) + [special[n] for n in rng.sample(names, 5)]
        _check(tok, s, cuts, some, f"soup {trial}")


def test_repeated_calls_and_changing_allowed_sets(tok):
    _, _, special = H.llama4()
    x, o = td_corpus.chat(1 << 20, seed=9)
    a_all = list(special.values())
    a_eot = [special["<|eot|>"]]
    r1 = _device(tok, x.tobytes(), o, a_all)
    r2 = _device(tok, x.tobytes(), o, a_eot)
    r3 = _device(tok, x.tobytes(), o, a_all)
    assert np.array_equal(r1[0], r3[0]) and np.array_equal(r1[1], r3[1])
    assert len(r2[0]) > len(r1[0])  # (the other specials are spelled out as ordinary text)
    plain = tok.encode_batch(x.tobytes(), o)
    r0 = _device(tok, x.tobytes(), o, [])
    assert np.array_equal(plain[0], r0[0]) and np.array_equal(plain[1], r0[1])


def test_text_that_is_mostly_candidates(tok):
    """Every other byte starts "<|": the candidate list of a workgroup overflows into the global one, clusters are long."""
    _, _, special = H.llama4()
    allowed = list(special.values())
    dense = (b"<|" * 3000 + b"<|eot|>" * 500 + b"<|e<|eot|>ot|>" * 300 + b"x<|eom|>" * 400)
    _check(tok, dense, [0, len(dense)], allowed, "dense candidates")
    _check(tok, dense, [0, 100, 101, 6001, len(dense)], allowed, "dense candidates, cut into documents")
