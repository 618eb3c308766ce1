"""Allowed special tokens pinned against the COMPILED REFERENCE where its behaviour is defined.

CoreBPE::encode(text, allowed_special) (/root/reference/src/tiktoken/tiktoken.cpp:169-234) looks for the next allowed special
with find_next_special_token (:130-154), which erases entries of the hash map it is iterating over (:143): with TWO or more
allowed specials the entry moved into the erased seat is skipped and its occurrences are tokenized as ordinary text (probed
here: 219 of 400 random two-special cases differ from tiktoken's semantics).  With ONE allowed special nothing can be skipped:
there the reference is deterministic, and it is exactly tiktoken's rule — cut at every occurrence, left to right, the text in
between encoded on its own.  These tests hold the restatement the other special-token tests use (and the product, on the GPU)
to the reference itself on that subset."""
import random

import numpy as np
import pytest

import helpers as H
from oracle import ref
from test_python_api import _tiktoken_special_split

NAMES = ["<|begin_of_text|>", "<|eot|>", "<|header_start|>", "<|text_post_train_reserved_special_token_7|>"]
FRAGS = ["hello", " world", "\n", "x = 1;", " naïve", " 中文", "", " ", "<|", "|>", "<|eot", "<|eot|", "<|begin_of_text|", "|><|", "<|eot|><|eot|>"]


def _cases(n, seed):
    rng = random.Random(seed)
    for _ in range(n):
        name = rng.choice(NAMES)
        parts = []
        for _ in range(rng.randrange(0, 10)):
            parts.append(rng.choice(FRAGS))
            if rng.random() < 0.5:
                parts.append(rng.choice(NAMES))  # (the allowed one, or another special's literal: ordinary text then)
        yield name, "".join(parts).encode("utf-8")
    for name in NAMES:  # edges: only the special, at both ends, back to back, cut short at the end
        for text in (name, name * 3, name + "a", "a" + name, name[:-1], "", "a" + name[:-1] + name):
            yield name, text.encode("utf-8")


@pytest.mark.skipif(not ref.available(), reason="compiled reference (oracle/_ref) not built")
def test_restatement_equals_the_reference_with_one_allowed_special():
    pat, mr, special = H.llama4()
    R = ref.RefTokenizer(pat, mr, special)
    n = 0
    for name, text in _cases(600, 11):
        want = R.encode_special(text, [name]).tolist()
        got = []
        for seg, sid in _tiktoken_special_split(text, {name.encode("utf-8"): special[name]}):
            got += R.encode(seg).tolist()
            if sid is not None:
                got.append(sid)
        assert got == want, (name, text)
        n += 1
    assert n > 600


@pytest.mark.gpu
@pytest.mark.skipif(not ref.available(), reason="compiled reference (oracle/_ref) not built")
def test_gpu_one_allowed_special_equals_the_reference():
    """The product through the C ABI (host search for small batches, td_special.hip for the large one) against
    CoreBPE::encode(text, {special}) of the compiled reference, document by document."""
    import td_corpus
    from tokendagger_amd import capi
    pat, mr, special = H.llama4()
    R = ref.RefTokenizer(pat, mr, special)
    tok = capi.HipTokenizer(pat, mr, special, device=0)
    by_name = {}
    for name, text in _cases(600, 12):
        by_name.setdefault(name, []).append(text)
    for name, docs in by_name.items():
        blob, offs = H.pack_docs(docs)
        toks, toffs = tok.encode_batch_with_special_strs(blob, offs, [name])
        for i, d in enumerate(docs):
            assert toks[toffs[i]:toffs[i + 1]].tolist() == R.encode_special(d, [name]).tolist(), (name, d)
    # one batch large enough for the device-side search (>= 1 MiB): chat-formatted text, ONE of its specials allowed
    x, offs = td_corpus.chat(3 << 20, seed=5)
    text = x.tobytes()
    for name in ("<|eot|>", "<|header_start|>"):
        toks, toffs = tok.encode_batch_with_special_strs(text, offs, [name])
        for d in range(0, len(offs) - 1, 7):
            doc = text[offs[d]:offs[d + 1]]
            assert toks[toffs[d]:toffs[d + 1]].tolist() == R.encode_special(doc, [name]).tolist(), (name, d)
    tok.close()
