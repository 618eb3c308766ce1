"""C++ vocabulary loaders (td_vocab_*, tokendagger_amd/csrc/td_vocab.cpp) against Python's own json / base64 on
files written by the test: tiktoken .model, Hugging Face tokenizer_config.json, tekken.json, the reference
wrapper's JSON files (formats: /root/reference/src/main.cpp:70-137, tests/throughput_test.py:106-180,
tokendagger/wrapper.py:116-134).  Host only: no GPU."""
import base64
import json
import random

import pytest

import helpers as H


@pytest.fixture(scope="module")
def capi():
    import __graft_entry__ as g
    g.build_hip()
    from tokendagger_amd import capi
    return capi


def _write_model(path, ranks: dict, newline="\n", trailing=""):
    with open(path, "w", newline="") as f:
        for b, r in ranks.items():
            f.write(base64.b64encode(b).decode() + " " + str(r) + trailing + newline)


def test_tiktoken_model_full_llama4(capi, tmp_path):
    pat, mr, sp = H.llama4()
    regular = {b: r for b, r in mr.items() if r < 200000}
    _write_model(tmp_path / "tokenizer.model", regular)
    cfg = {"added_tokens_decoder": {str(i): {"content": s, "lstrip": False, "special": True} for s, i in sp.items()},
           "bos_token": "<|begin_of_text|>", "model_max_length": 10485760, "clean_up_tokenization_spaces": False}
    (tmp_path / "tokenizer_config.json").write_text(json.dumps(cfg, indent=2))
    v = capi.Vocab().load_tiktoken(tmp_path / "tokenizer.model").load_hf_special(tmp_path / "tokenizer_config.json", True)
    assert v.mergeable_ranks() == mr          # the reference's tests put the specials into mergeable_ranks too
    assert v.special_tokens() == sp
    assert capi.load_tiktoken_bpe(tmp_path / "tokenizer.model") == regular
    v2 = capi.Vocab().load_tiktoken(tmp_path / "tokenizer.model").load_hf_special(tmp_path / "tokenizer_config.json", False)
    assert v2.mergeable_ranks() == regular and v2.special_tokens() == sp


def test_model_file_line_forms(capi, tmp_path):
    ranks = {b"a": 0, b"\x00\xff": 1, b" the": 7, bytes(range(256)): 300000}
    _write_model(tmp_path / "crlf.model", ranks, newline="\r\n", trailing="  ")
    assert capi.load_tiktoken_bpe(tmp_path / "crlf.model") == ranks
    (tmp_path / "blank.model").write_text("\n\nYQ== 0\n   \nYg==\t1")  # blank lines, tab separator, no final newline
    assert capi.load_tiktoken_bpe(tmp_path / "blank.model") == {b"a": 0, b"b": 1}
    for name, body, what in [("b64", "Y!== 0\n", "base64"), ("rank", "YQ== x\n", "rank"), ("neg", "YQ== -1\n", "rank"),
                             ("one", "YQ==\n", "expected"), ("junk", "YQ== 1 2\n", "rank")]:
        (tmp_path / name).write_text("Yg== 5\n" + body)
        with pytest.raises(capi.TokenDaggerHipError) as e:
            capi.load_tiktoken_bpe(tmp_path / name)
        assert e.value.code == 3 and what in str(e.value) and ":2:" in str(e.value)
    with pytest.raises(capi.TokenDaggerHipError) as e:
        capi.load_tiktoken_bpe(tmp_path / "does_not_exist.model")
    assert "cannot open" in str(e.value)


def test_hf_config_json_escapes(capi, tmp_path):
    # JSON string escapes incl. \u surrogate pairs must come out as the UTF-8 Python produces
    specials = {"<|a|>": 5, "tab\there": 6, "quote\"q\\": 7, "snow☃": 8, "emoji\U0001F600!": 9, "bell\x07x": 10, "sl/ash": 11}
    doc = {"x": [1, -2.5e3, True, None, {"deep": [[], {}]}],
           "added_tokens_decoder": {str(i): {"special": True, "content": s} for s, i in specials.items()}}
    for ensure_ascii in (True, False):
        (tmp_path / "c.json").write_text(json.dumps(doc, ensure_ascii=ensure_ascii), encoding="utf-8")
        assert capi.Vocab().load_hf_special(tmp_path / "c.json").special_tokens() == specials
    (tmp_path / "none.json").write_text('{"model_max_length": 1}')
    assert capi.Vocab().load_hf_special(tmp_path / "none.json").special_tokens() == {}
    for name, body in [("trunc", '{"added_tokens_decoder": {"1": {"content": "a"}'), ("key", '{"added_tokens_decoder": {"x": {"content": "a"}}}'),
                       ("nocontent", '{"added_tokens_decoder": {"1": {"special": true}}}'), ("trail", '{} x'), ("esc", '{"a": "\\q"}')]:
        (tmp_path / name).write_text(body)
        with pytest.raises(capi.TokenDaggerHipError) as e:
            capi.Vocab().load_hf_special(tmp_path / name)
        assert e.value.code == 3 and name in str(e.value)


def test_json_reader_fuzz_against_python(capi, tmp_path):
    rng = random.Random(11)
    alphabet = ["a", "é", "中", "\U0001F680", "\\", "\"", "\n", "\t", "\x01", "/", " ", " ", "\x7f"]

    def rnd_str():
        return "".join(rng.choice(alphabet) for _ in range(rng.randint(1, 8)))

    for it in range(40):
        specials = {}
        while len(specials) < 20:
            specials[rnd_str()] = rng.randint(0, 2 ** 31 - 1)
        doc = {"pad": [rnd_str() for _ in range(5)],
               "added_tokens_decoder": {str(i): {"content": s, "n": rng.random()} for s, i in specials.items()}}
        (tmp_path / "f.json").write_text(json.dumps(doc, ensure_ascii=bool(it & 1), indent=rng.choice([None, 1])), encoding="utf-8")
        got = capi.Vocab().load_hf_special(tmp_path / "f.json").special_tokens()
        assert got == {s: i for s, i in specials.items()}


def test_tekken_layout(capi, tmp_path):
    rng = random.Random(3)
    n_special, n_total = 10, 300
    toks = [bytes([i]) for i in range(256)] + [bytes(rng.randrange(256) for _ in range(rng.randint(2, 12))) for _ in range(100)]
    pat = r"[^\r\n\p{L}\p{N}]?[\p{Lu}\p{Lt}\p{Lm}\p{Lo}\p{M}]*[\p{Ll}\p{Lm}\p{Lo}\p{M}]+|\p{N}| ?[^\s\p{L}\p{N}]+[\r\n/]*|\s*[\r\n]+|\s+(?!\S)|\s+"
    doc = {"config": {"pattern": pat, "num_vocab_tokens": len(toks), "default_vocab_size": n_total,
                      "default_num_special_tokens": n_special, "version": "v7"},
           "vocab": [{"rank": i, "token_bytes": base64.b64encode(t).decode(), "token_str": t.decode("utf-8", "replace") if i % 3 else None}
                     for i, t in enumerate(toks)],
           "special_tokens": [{"rank": i, "token_str": f"<s{i}>", "is_control": True} for i in range(n_special)]}
    (tmp_path / "tekken.json").write_text(json.dumps(doc))
    v = capi.Vocab().load_tekken(tmp_path / "tekken.json")
    # reference loader (tests/throughput_test.py:121-131): the first default_vocab_size - default_num_special_tokens
    # entries, id = index + default_num_special_tokens, no special tokens
    blob, offs, ranks = v.arrays()
    assert len(ranks) == n_total - n_special and ranks.tolist() == list(range(n_special, n_total))
    raw = blob.tobytes()
    assert [raw[offs[i]:offs[i + 1]] for i in range(len(ranks))] == toks[:n_total - n_special]
    assert v.pattern == pat and v.special_tokens() == {}
    doc["config"]["default_vocab_size"] = 10 ** 6
    (tmp_path / "short.json").write_text(json.dumps(doc))
    with pytest.raises(capi.TokenDaggerHipError):
        capi.Vocab().load_tekken(tmp_path / "short.json")


def test_wrapper_json_files(capi, tmp_path):
    vocab = [{"rank": i, "token_bytes": list(bytes([i])), "token_string": ""} for i in range(256)]
    vocab += [{"rank": 256, "token_bytes": [104, 105]}, {"rank": 257, "token_bytes": list("é".encode()), "token_string": "é"}]
    special = {"<|endoftext|>": 1000, "<|fim|>": 1001}
    (tmp_path / "vocab.json").write_text(json.dumps(vocab))
    (tmp_path / "special.json").write_text(json.dumps(special))
    v = capi.Vocab().load_json(tmp_path / "vocab.json", tmp_path / "special.json")
    assert v.mergeable_ranks() == {bytes(e["token_bytes"]): e["rank"] for e in vocab}
    assert v.special_tokens() == special
    v.set_pattern("abc")
    assert v.pattern == "abc"
    (tmp_path / "bad.json").write_text(json.dumps([{"rank": 1, "token_bytes": [300]}]))
    with pytest.raises(capi.TokenDaggerHipError):
        capi.Vocab().load_json(tmp_path / "bad.json")


def test_base64_matches_python(capi, tmp_path):
    rng = random.Random(5)
    ranks = {}
    for r in range(2000):
        ranks[bytes(rng.randrange(256) for _ in range(rng.randint(1, 40))) + r.to_bytes(3, "big")] = r
    _write_model(tmp_path / "r.model", ranks)
    assert capi.load_tiktoken_bpe(tmp_path / "r.model") == ranks
