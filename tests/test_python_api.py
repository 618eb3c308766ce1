"""Drop-in Python surface (tokendagger_amd.Tokenizer / Encoding, `import tokendagger as tiktoken`) on the GPU."""
import numpy as np
import pytest

import cases
import helpers as H

pytestmark = pytest.mark.gpu

# a pattern that is really outside the generic subset (back-reference + lazy quantifier), shared with the CPU-side contract
# test (tests/test_abi.py::test_encoding_constructor_rejects_unsupported_pattern_on_cpu): the reference's constructor cannot
# build a usable tokenizer from a pattern it cannot compile either (tiktoken.cpp:59-62)
BAD_PATTERN = cases.UNSUPPORTED_PATTERN


@pytest.fixture(scope="module")
def enc():
    import tokendagger as tiktoken
    pat, mr, special = H.llama4()
    return tiktoken.Encoding(name="llama4", pat_str=pat, mergeable_ranks=mr, special_tokens=special)


def test_encode_matches_golden_strings(enc, golden):
    text, offs = golden["text"].tobytes(), golden["offsets"]
    gold, go = golden["enc"], golden["enc_offsets"]
    names = list(golden["names"])
    for name, s in cases.all_strings():
        d = names.index(name)
        exp = gold[go[d]:go[d + 1]].tolist()
        assert enc.encode(s) == exp, name
        assert enc.encode_ordinary(s) == exp, name
        assert enc.encode(s, allowed_special=set(), disallowed_special=set()) == exp


def test_batch_and_numpy_forms(enc):
    O = H.port_tokenizer()
    texts = [s for _, s in cases.all_strings()]
    got = enc.encode_batch(texts, num_threads=4)
    assert got == [O.encode(t.encode("utf-8")).tolist() for t in texts]
    assert enc.encode_ordinary_batch(texts[:50]) == got[:50]
    arr = enc.encode_to_numpy("Hello, world!")
    assert arr.dtype == np.int32 and arr.tolist() == [19873, 24, 3817, 13]
    blob, offs = H.pack_docs([t.encode("utf-8") for t in texts])
    toks, toffs = enc.encode_batch_to_numpy(blob, offs)
    assert toks.tolist() == [x for g in got for x in g] and toffs[-1] == len(toks)


def test_decode_and_roundtrip(enc, golden):
    for ids, exp in zip(golden["decode_ids"], golden["decode_bytes"]):
        assert enc.decode_bytes(list(ids)) == exp
    for s in ["Hello, world!", "The quick brown fox jumps over the lazy dog.", "Unicode: 你好 \U0001F30D"]:
        assert enc.decode(enc.encode(s)) == s
    assert enc.decode_batch([[19873, 24, 3817, 13], [220]]) == ["Hello, world!", " "]
    assert enc.decode_single_token_bytes(220) == b" "
    import tokendagger
    with pytest.raises(tokendagger.TokenDaggerError):
        enc.decode([10 ** 8])


def test_special_tokens_tiktoken_semantics(enc):
    O = H.port_tokenizer()
    _, _, special = H.llama4()
    bos, eos = special["<|begin_of_text|>"], special["<|end_of_text|>"]
    s = "<|begin_of_text|>Hello<|end_of_text|> tail <|begin_of_text|>"
    exp = [bos] + O.encode(b"Hello").tolist() + [eos] + O.encode(b" tail ").tolist() + [bos]
    assert enc.encode(s, allowed_special="all") == exp
    assert enc.encode(s, allowed_special={"<|begin_of_text|>", "<|end_of_text|>"}) == exp
    assert enc.encode_with_special_tokens(s) == exp
    only_eos = O.encode(b"<|begin_of_text|>Hello").tolist() + [eos] + O.encode(b" tail <|begin_of_text|>").tolist()
    assert enc.encode(s, allowed_special={"<|end_of_text|>"}) == only_eos
    assert enc.encode(s) == O.encode(s.encode()).tolist(), "default: specials are ordinary text"
    with pytest.raises(ValueError):
        enc.encode(s, disallowed_special="all")
    import tokendagger
    with pytest.raises(tokendagger.TokenDaggerError):
        enc.encode("x", allowed_special={"<|no_such_token|>"})
    # second element of CoreBPE.encode's return pair: ids of the last regex piece (0 after a special)
    toks, last = enc._core_bpe.encode("Hello world", set())
    assert toks == O.encode(b"Hello world").tolist() and last == 1
    assert enc._core_bpe.encode(s, {"<|begin_of_text|>"})[1] == 0
    assert enc._core_bpe.encode("abc \U0001F468‍\U0001F4BB", set())[1] == len(O.encode("\U0001F468‍\U0001F4BB".encode()))


def test_attributes_and_errors(enc):
    import tokendagger
    assert enc.n_vocab == 201134 and enc.max_token_value == 201133
    assert enc.is_special_token(200000) and not enc.is_special_token(5)
    assert "<|begin_of_text|>" in enc.special_tokens_set and len(enc.special_tokens()) == 1134
    assert repr(enc) == "<TokenDagger 'llama4'>"
    with pytest.raises(tokendagger.TokenDaggerError):
        tokendagger.Encoding(name="bad", pat_str=BAD_PATTERN, mergeable_ranks={b"a": 0})
    t = tokendagger.create_tokenizer("toy", enc.pattern, [{"rank": 0, "token_bytes": [97]}, {"rank": 1, "token_bytes": [98]},
                                                           {"rank": 2, "token_bytes": [97, 98], "token_string": "ab"}])
    assert t.encode("ab") == [2] and t.encode("ba") == [1, 0]
    with pytest.raises(tokendagger.TokenDaggerError):
        t.encode("abc")


def test_from_files_matches_in_memory_construction(enc, tmp_path):
    """C++ loader path (Tokenizer.from_files / load_tokenizer) builds the same tokenizer as Encoding(...)."""
    import base64
    import json
    import tokendagger as tiktoken
    pat, mr, sp = H.llama4()
    with open(tmp_path / "tokenizer.model", "w") as f:
        for b, r in mr.items():
            if r < 200000:
                f.write(base64.b64encode(b).decode() + " " + str(r) + "\n")
    (tmp_path / "tokenizer_config.json").write_text(json.dumps(
        {"added_tokens_decoder": {str(i): {"content": s} for s, i in sp.items()}}))
    t = tiktoken.Tokenizer.from_files("llama4-files", pat_str=pat, tiktoken_model=tmp_path / "tokenizer.model",
                                      hf_config=tmp_path / "tokenizer_config.json", specials_mergeable=True)
    assert t.pattern == pat and t.max_token_value == enc.max_token_value and t.n_vocab == enc.n_vocab
    assert t.special_tokens_set == enc.special_tokens_set
    for s in ["Hello, world!", "def f(x):\n    return x**2  # 中文 😀", "<|begin_of_text|>hi<|eot|>", " " * 70 + "end"]:
        assert t.encode(s) == enc.encode(s)
        assert t.encode(s, allowed_special="all") == enc.encode(s, allowed_special="all")
        assert t.decode(t.encode(s)) == s
    assert tiktoken.load_tiktoken_bpe(tmp_path / "tokenizer.model") == {b: r for b, r in mr.items() if r < 200000}
    # the reference wrapper's JSON formats through load_tokenizer (wrapper.py:333-355)
    small = [{"rank": r, "token_bytes": list(b), "token_string": ""} for b, r in mr.items() if r < 3000]
    (tmp_path / "vocab.json").write_text(json.dumps(small))
    (tmp_path / "special.json").write_text(json.dumps({"<|x|>": 5000}))
    lt = tiktoken.load_tokenizer("small", tmp_path / "vocab.json", pat, tmp_path / "special.json")
    ref_small = tiktoken.Encoding("small", pat_str=pat, mergeable_ranks={bytes(e["token_bytes"]): e["rank"] for e in small},
                                  special_tokens={"<|x|>": 5000})
    for s in ["the quick brown fox", "a<|x|>b"]:
        assert lt.encode(s, allowed_special="all") == ref_small.encode(s, allowed_special="all")
    with pytest.raises(FileNotFoundError):
        tiktoken.load_tokenizer("none", tmp_path / "missing.json", pat)
    with pytest.raises(tiktoken.TokenDaggerError):
        tiktoken.Tokenizer.from_files("bad", pat_str=BAD_PATTERN, tiktoken_model=tmp_path / "tokenizer.model")


def test_single_token_accessors(enc):
    pat, mr, special = H.llama4()
    assert enc.encode_single_token("hello") == mr[b"hello"] and enc.encode_single_token(b" the") == mr[b" the"]
    assert enc.encode_single_token("<|begin_of_text|>") == special["<|begin_of_text|>"]
    for bad in ["hello world, this is not one token", b"\xff\xfe\xfd\xfc\xfb\xfa"]:
        with pytest.raises(KeyError):
            enc.encode_single_token(bad)
    assert enc.decode_single_token_bytes(mr[b"hello"]) == b"hello"
    assert enc.decode_tokens_bytes([19873, 24, 3817, 13]) == [b"Hello", b",", b" world", b"!"]
    with pytest.raises(KeyError):
        enc.decode_single_token_bytes(enc.n_vocab + 5)
    tbv = enc.token_byte_values()
    assert len(tbv) == len(set(mr.values()) - set(special.values())) and tbv == sorted(tbv) and b"hello" in tbv
    with pytest.raises(KeyError):
        enc.eot_token  # Llama-4 names its end token differently: same behaviour as a tiktoken Encoding without "<|endoftext|>"


def test_decode_batch_one_pass(enc):
    docs = ["Hello, world!", "", "中文 😀\n", "x" * 5000, "the quick brown fox " * 300, ""]
    ids = enc.encode_batch(docs)
    assert enc.decode_batch(ids) == docs
    assert enc.decode_bytes_batch(ids) == [d.encode("utf-8") for d in docs]
    assert enc.decode_batch([]) == [] and enc.decode_batch([[]]) == [""]
    import tokendagger as tiktoken
    with pytest.raises(tiktoken.TokenDaggerError) as e:
        enc.decode_batch([[1, 2], [3, 10 ** 8]])
    assert "Invalid token for decoding: 100000000" in str(e.value)
    # through the C ABI with numpy arrays
    from tokendagger_amd import capi
    pat, mr, special = H.llama4()
    tok = capi.HipTokenizer(pat, mr, special, device=0)
    flat = np.asarray([t for d in ids for t in d], dtype=np.int32)
    offs = np.cumsum([0] + [len(d) for d in ids]).astype(np.int64)
    blob, boffs = tok.decode_batch(flat, offs)
    assert [blob[boffs[i]:boffs[i + 1]] for i in range(len(docs))] == [d.encode("utf-8") for d in docs]


def test_encode_batch_with_allowed_special_is_one_batch(enc):
    texts = ["<|begin_of_text|>Hello<|eot|>", "", "no specials here", "<|eot|><|eot|>x", "a<|not_a_token|>b",
             "<|begin_of_text|>" * 50 + "tail", "edge <|e", "<|header_start|>user<|header_end|>\n\nHi<|eot|>" * 200]
    got = enc.encode_batch(texts, allowed_special="all")
    assert got == [enc.encode(t, allowed_special="all") for t in texts]
    some = {"<|eot|>"}
    assert enc.encode_batch(texts, allowed_special=some) == [enc.encode(t, allowed_special=some) for t in texts]
    # a large document with every special allowed: one pass over the text, not one search per special
    big = ("lorem ipsum dolor sit amet <tag> x < y " * 40000) + "<|eot|>"
    ids = enc.encode(big, allowed_special="all")
    assert ids[-1] == enc._special_tokens["<|eot|>"] and enc.decode(ids) == big


def test_from_files_tekken_json(tmp_path):
    """tekken.json through the C++ loader (pattern from the file, id = index + default_num_special_tokens) builds the
    same tokenizer as Encoding(...) over the same entries."""
    import base64
    import json
    import tokendagger as tiktoken
    from tokendagger_amd import vocab_io
    _, mr, _ = H.llama4()
    by_rank = sorted(((r, b) for b, r in mr.items() if r < 30000), key=lambda x: x[0])
    n_special = 1000
    doc = {"config": {"pattern": vocab_io.TEKKEN_PAT_STR, "default_vocab_size": len(by_rank) + n_special,
                      "default_num_special_tokens": n_special, "version": "v7-synthetic"},
           "vocab": [{"rank": i, "token_bytes": base64.b64encode(b).decode(), "token_str": None} for i, (_, b) in enumerate(by_rank)] +
                    [{"rank": 10 ** 6, "token_bytes": base64.b64encode(b"never loaded").decode(), "token_str": "x"}]}
    (tmp_path / "tekken.json").write_text(json.dumps(doc))
    t = tiktoken.Tokenizer.from_files("tekken-synthetic", tekken=tmp_path / "tekken.json")
    ref = tiktoken.Encoding("same", pat_str=vocab_io.TEKKEN_PAT_STR, mergeable_ranks={b: i + n_special for i, (_, b) in enumerate(by_rank)})
    assert t.pattern == vocab_io.TEKKEN_PAT_STR and t.max_token_value == ref.max_token_value
    for s in ["Hello, world! It's 2024.", "def f(x):\n    return x**2  # 中文 😀", "12345 67", "  spaces   and\ttabs\n\n"]:
        ids = t.encode(s)
        assert ids == ref.encode(s) and min(ids) >= n_special and t.decode(ids) == s


# ---- special tokens: property test against a restatement of tiktoken's splitting ------------------------------------
def _tiktoken_special_split(data: bytes, allowed: dict[bytes, int]):
    """tiktoken's encode() with allowed_special, restated (tiktoken: `_encode_native` / CoreBPE.encode in the reference,
    tiktoken.cpp:169-234 minus its iterator bug): scan for the EARLIEST position where an allowed special
    string occurs; the text before it is ordinary text, the special contributes its id, continue behind it.  Several
    allowed strings at one position: tiktoken's alternation order is a hash-map order, i.e. unspecified; this
    implementation defines it as the LONGEST one.  -> [(ordinary bytes, special id or None)]"""
    out, start, p = [], 0, 0
    while p < len(data):
        best = None
        for s in allowed:
            if data.startswith(s, p) and (best is None or len(s) > len(best)):
                best = s
        if best is None:
            p += 1
            continue
        out.append((data[start:p], allowed[best]))
        p += len(best)
        start = p
    out.append((data[start:], None))
    return out


def test_special_splitting_property():
    from hypothesis import given, settings, strategies as st, HealthCheck
    from tokendagger_amd import capi
    pat, mr, _ = H.llama4()
    ranks = {b: r for b, r in mr.items() if r < 200000}
    # overlapping / nested / prefix-sharing specials, one pair sharing an id, one single-byte special
    specials = {"<|a|>": 200000, "<|a|><|b|>": 200001, "<|b|>": 200002, "<|": 200003, "|>": 200004, "<|ab|>": 200005,
                "<|a|>x": 200006, "\u00a7": 200007, "<|alias1|>": 200008, "<|alias2|>": 200008, "<|a": 200009}
    tok = capi.HipTokenizer(pat, ranks, specials, device=0)
    O = H.port_tokenizer()
    sp_bytes = {k.encode("utf-8"): v for k, v in specials.items()}
    frag = st.sampled_from(list(specials) + ["<", "|", ">", "a", "b", "x", " ", "hello", "\n", "<|a|", "|><|", "\u00a7\u00a7", "wörld", "日本"])
    texts = st.lists(frag, min_size=0, max_size=24).map("".join)
    subsets = st.sets(st.sampled_from(list(specials)), max_size=len(specials))

    @settings(max_examples=150, deadline=None, suppress_health_check=list(HealthCheck))
    @given(st.lists(texts, min_size=1, max_size=6), subsets)
    def run(docs, allowed):
        allowed_b = {k.encode("utf-8"): specials[k] for k in allowed}
        exp = []
        for d in docs:
            ids = []
            for seg, sid in _tiktoken_special_split(d.encode("utf-8"), allowed_b):
                ids += O.encode(seg).tolist()
                if sid is not None:
                    ids.append(sid)
            exp.append(ids)
        blob, offs = H.pack_docs([d.encode("utf-8") for d in docs])
        toks, toffs = tok.encode_batch_with_special_strs(blob, offs, sorted(allowed))
        got = [toks[toffs[i]:toffs[i + 1]].tolist() for i in range(len(docs))]
        assert got == exp, (docs, allowed)
        one, _ = tok.encode_with_special_strs(docs[0].encode("utf-8"), sorted(allowed))
        assert one.tolist() == exp[0]

    run()
    # ids instead of strings: every string carrying an allowed id is cut out (alias1 and alias2 share 200008)
    t, _ = tok.encode_with_special(b"x<|alias1|>y<|alias2|>z", [200008])
    assert t.tolist() == O.encode(b"x").tolist() + [200008] + O.encode(b"y").tolist() + [200008] + O.encode(b"z").tolist()
    t, _ = tok.encode_with_special_strs(b"x<|alias1|>y<|alias2|>z", ["<|alias1|>"])
    assert t.tolist() == O.encode(b"x").tolist() + [200008] + O.encode(b"y<|alias2|>z").tolist()
    with pytest.raises(capi.TokenDaggerHipError):
        tok.encode_with_special_strs(b"x", ["<|nope|>"])
    tok.close()


def test_calls_leave_the_current_device_alone():
    """Every entry point restores the caller's current HIP device (a tokenizer on another GPU must not
    re-point torch); with one GPU: the device is 0 before and after, and last-error text is per thread."""
    import threading
    import torch
    from tokendagger_amd import capi
    pat, mr, sp = H.llama4()
    tok = capi.HipTokenizer(pat, mr, sp, device=0)
    assert torch.cuda.current_device() == 0
    tok.encode(b"hello")
    assert torch.cuda.current_device() == 0
    errs = []

    def bad():
        try:
            tok.decode_bytes([10 ** 8])
        except capi.TokenDaggerHipError as e:
            errs.append(str(e))

    th = [threading.Thread(target=bad) for _ in range(4)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert len(errs) == 4 and all("Invalid token for decoding: 100000000" in e for e in errs)
    tok.close()


def test_encode_batch_lists_from_several_threads(enc):
    """`encode_batch(list[str]) -> list[list[int]]` builds its result with shared int objects, reference counts added per distinct
    id and the lists' item arrays filled by threads with the GIL released (csrc/py_binding.cpp: IntCache; the reference fans
    single calls out to a thread pool, tokendagger/wrapper.py:212-235).  Several Python threads on ONE Encoding at once: the same
    lists as one after the other, ordinary mutable lists of ints, and the shared ints' reference counts back to where they
    were when the lists are gone."""
    import gc
    import sys
    import threading
    import td_corpus
    texts = []
    for seed in range(4):
        x, offs = td_corpus.english(3 << 20, seed=40 + seed)
        s = x.tobytes().decode("ascii")
        texts.append([s[offs[i]:offs[i + 1]] for i in range(0, len(offs) - 1)] + ["", "x", "naïve 中文 <|eot|>"])
    want = [enc.encode_batch(t) for t in texts]
    for w, t in zip(want, texts):
        assert type(w) is list and len(w) == len(t) and type(w[0]) is list and type(w[0][0]) is int
        assert w[1] == enc.encode(t[1]) and w[-1] == enc.encode(t[-1]) and w[-3] == []
    probe = want[0][0][0]
    before = sys.getrefcount(probe)
    got = [None] * 4
    errors = []

    def work(k):
        try:
            for _ in range(3):
                got[k] = enc.encode_batch(texts[k])
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    assert got == want
    got[0][0].append(7)  # (ordinary lists)
    del got
    gc.collect()
    assert sys.getrefcount(probe) == before
