"""Generic split patterns: td_regex.cpp compiles the backtracking subset of PCRE2 syntax tokenizer patterns are
written in, td_regex.h matches it — on the device one lane per document (td_generic.hip).  Pinned against PCRE2 itself: the
compiled reference runs ANY pattern (tiktoken.cpp:47-128), so pieces (with the text a pattern skips) and ids are compared
with it directly.  CPU: compiler + matcher (the same header the kernel compiles); GPU: ids through the C ABI."""
import random
import zlib

import numpy as np
import pytest

import helpers as H
from oracle import ref
from tokendagger_amd import vocab_io

AUTOGEN = r"[a-zA-Z]+|\s+|[0-9]+|[^\w\s]"          # the reference's tests/autogenned_test.py:66 (skips '_' and 'é')
PATTERNS = {
    "autogen": AUTOGEN,
    # every member of the family through the GENERIC compiler (the product runs them on their own kernels)
    "llama4": vocab_io.LLAMA4_PAT_STR, "tekken": vocab_io.TEKKEN_PAT_STR, "cl100k": vocab_io.CL100K_PAT_STR,
    "cl100k_possessive": vocab_io.CL100K_PAT_STR_POSSESSIVE, "cl100k_current": vocab_io.CL100K_PAT_STR_CURRENT,
    "qwen2": vocab_io.QWEN2_PAT_STR, "gpt2": vocab_io.GPT2_PAT_STR, "gpt2_possessive": vocab_io.GPT2_PAT_STR_POSSESSIVE,
    # other shapes: \w \d classes, bounded repeats, ranges, dots, look-aheads, literals, skipped text, a tail that never matches
    "words": r"\w+|[^\w\s]+|\s+", "digits": r"\d{1,3}|\D+", "camel": r"[A-Z][a-z]*|\s|.", "cats": r"\p{Lu}\p{Ll}*|\p{Nd}+|\.{2,}|.",
    "look": r"[a-z]+(?=[0-9])|[0-9]+|\s+(?!\S)|\S", "letters_only": r"\p{L}+", "lit": r"(?i:the|an|a|k|s)| ?\pL+|\PL",
    "wordb": r"\b\w+\b|\B[-_]+\B|\S", "anchors": r"^\s+|\A[#]+|\w+\z|\w+|\s+\Z|\W", "bnd2": r"\Bs\b|[a-z]+?"[:-1] + r"|.",
    "scripts": r"\p{Han}+|[\p{Hiragana}\p{Katakana}ー]+|\p{Hangul}+|\p{Latin}+|\P{Thai}|\s+|.", "cyr": r"[\p{Cyrillic}\p{Greek}]+\d*|\p{Any}",
    "hex": r"0x[0-9a-fA-F]{1,8}|\x41+|[\x{4e00}-\x{9fff}]+|[^\S\n]*\n|.", "opt": r"(?:ab|a)?c|[ab]+|\s*+x|.",
    # atomic optional groups: once a literal is chosen the matcher never comes back for the next one or the skip
    "atomic1": r"(?:a|ab)?+c|.", "atomic2": r"(?:ab|a)?+bc|.", "atomic3": r" ?(?:the|a|an)?+[a-z]+|\s+|.",
    # lazy quantifiers: as few characters as possible first, one more per way back (PCRE2_NOTEMPTY makes a lone "x*?" take one)
    "lazy1": r"[a-z]+?[0-9]|\s*?\S|.", "lazy2": r"\w{2,5}?\b|\w+?|\s??x|.", "lazy3": r"[ab]*?c|(?:ab|a)??b+|\p{L}*?\p{Lu}|\s+?", "lazy4": r"a??b|.{1,3}?[.!]|\d+?(?=\d)|.",
    # look-behinds of one character class
    # repeated groups of single-character alternatives (= repeated classes)
    "rep1": r"(?:a|b|[cd])+x?|(?:é|ü|0)*[.]|\s+|.", "rep2": r"(?i:s|k|t){2,4}|(?:-|_)++\w|(?:a|b)*?c|.",
    # horizontal white space, "not a newline", POSIX classes (what PCRE2_UCP makes of them)
    "horiz": r"\h+|[^\h\n]+|\N", "horiz2": r"\H{1,3}|\h", "vert": r"\v+|[^\v\h]+|\V", "posix1": r"[[:alpha:]]+|[[:digit:]]+|[[:space:]]+|[^[:alnum:][:space:]]+", "posix2": r"[[:upper:]][[:lower:]]*|[[:word:]]+|[[:cntrl:]]|[[:^alpha:]]",
    "lookb1": r"(?<=[a-z])[0-9]+|(?<![0-9])[a-z]+|\s+|.", "lookb2": r"(?<!\S)\w+|(?<=\s)[^\w\s]+|\S|\s+(?<=\n)", "lookb3": r"\p{L}+(?<=s)|(?<=\p{Han})\p{Han}|(?<!.)#+|.",
}
REJECTED = [r"\012|.", r"[\d-z]", r"[\p{Han}-z]", r"\x{D800}", r"[\x{DFFF}]", r"(\w+)\s+\1", r"\b+x", r"\Gabc", r"(?<=ab)c", r"(?<=a)+b", r"(?<=a|b)c", r"[[:punct:]]+", r"[[:graph:]]", r"\R", r"(a|b)+", r"\p{Klingon}+", r"[^\P{Han}]", r"(?i)abc", r"a|", r"(?:ab|c)*", r"(?:a|)+"]


def _strings(n, seed):
    rng = random.Random(seed)
    for i in range(n):
        k = i % 4
        if k == 0:
            yield "".join(rng.choice(" \t\n\r_aAbBtThHeExX09zZ.,'\"(){}-+=é中ſK😀/") for _ in range(rng.randrange(0, 60)))
        elif k == 1:
            yield H.fuzz_string(rng, 60)
        elif k == 2:
            yield H.random_unicode_string(rng, 40)
        else:
            yield "".join(rng.choice(["the ", "An", " a", "0x1F", "ab", "abc", "c", " x", "12345", "\n\n", "  ", "HelloWorld", "...", "中文"]) for _ in range(rng.randrange(1, 12)))


def test_unsupported_syntax_is_rejected_with_a_reason():
    for pat in REJECTED:
        with pytest.raises(ValueError) as e:
            H.rx_split(pat, b"abc")
        assert "split pattern" in str(e.value) or "support" in str(e.value), (pat, str(e.value))


@pytest.mark.skipif(not ref.available(), reason="compiled reference (oracle/_ref) not built")
@pytest.mark.parametrize("name", sorted(PATTERNS))
def test_pieces_equal_pcre2(name, golden):
    pat = PATTERNS[name]
    _, mr, special = H.llama4()
    R = ref.RefTokenizer(pat, mr, special)
    for s in _strings(2500, zlib.crc32(name.encode()) & 0xFFFF):  # (deterministic: str hashes are salted per process)
        b = s.encode("utf-8")
        assert [b[a:e] for a, e in H.rx_split(pat, b)] == R.split_pieces(b), (name, s)
    text, offs = golden["text"].tobytes(), golden["offsets"]
    for d in range(0, len(offs) - 1, 41):
        doc = text[offs[d]:offs[d + 1]]
        if len(doc) > 20000:
            continue
        assert [doc[a:e] for a, e in H.rx_split(pat, doc)] == R.split_pieces(doc), (name, golden["names"][d])


@pytest.mark.skipif(not ref.available(), reason="compiled reference (oracle/_ref) not built")
def test_unbounded_quantifiers_have_no_65535_limit():
    """'*', '+' and '{m,}' stopped after 65535 characters (the bound was stored in 16 bits and used as a real one)."""
    _, mr, special = H.llama4()
    for pat, doc in ((r"[a-z]+|\s+", b"a" * 70000 + b" b"), (r"\s*[a-z]|\S", b" " * 66000 + b"x."), (r"[a-z]{2,}|.", b"q" * 131075 + b"!"),
                     (r"\p{L}+|.", "é".encode() * 66001 + b"1")):
        R = ref.RefTokenizer(pat, mr, special)
        assert [doc[a:e] for a, e in H.rx_split(pat, doc)] == R.split_pieces(doc), pat
    assert H.rx_split(r"(?:a|ab)?+c|.", b"abc") == [(0, 1), (1, 2), (2, 3)]
    assert H.rx_split(r"(?:ab|a)?+bc|.", b"abc") == [(0, 1), (1, 3)]


def test_patterns_beyond_the_program_limits_are_rejected_not_truncated():
    for pat in ("|".join("a%d" % i for i in range(40)),              # alternatives
                "".join("[a-%c]" % chr(ord("b") + i % 20) for i in range(20)),  # elements of one alternative
                "(?:" + "|".join("w%03d" % i for i in range(120)) + ")"):      # literals of a group
        with pytest.raises(ValueError) as e:
            H.rx_split(pat, b"abc")
        assert "too" in str(e.value), (pat[:30], str(e.value))
    # 31 alternatives are fine; ordered alternation: "a2" stands in front of "a29"
    assert H.rx_split("|".join("a%d" % i for i in range(30)) + "|.", b"a7a29x") == [(0, 2), (2, 4), (4, 5), (5, 6)]


def test_the_reference_tests_own_pattern_skips_text():
    spans = H.rx_split(AUTOGEN, "snake_case é 42!".encode())
    assert [(a, e) for a, e in spans] == [(0, 5), (6, 10), (10, 11), (13, 14), (14, 16), (16, 17)]  # '_' and 'é' are skipped


@pytest.mark.gpu
@pytest.mark.skipif(not ref.available(), reason="compiled reference (oracle/_ref) not built")
def test_gpu_generic_patterns_equal_the_reference():
    import td_corpus
    from tokendagger_amd import capi
    _, mr, special = H.llama4()
    for name in ("autogen", "words", "look", "letters_only", "opt", "atomic3", "lazy2", "lookb2"):
        pat = PATTERNS[name]
        tok = capi.HipTokenizer(pat, mr, special, device=0)
        R = ref.RefTokenizer(pat, mr, special)
        docs = [s.encode("utf-8") for s in _strings(400, 7 + len(name))]
        docs += [b"", b"_", b"___", "é".encode(), b"a_b", b"snake_case_name = 42", ("word_" * 3000).encode(), ("x" * 5000 + "_").encode()]
        if name in ("autogen", "words", "letters_only"):
            # runs longer than 65535 characters.  Only for patterns that match such a run in one go: "look" backs
            # out of [a-z]+(?=[0-9]) one character at a time at every start, quadratic for PCRE2 and for one GPU lane alike
            docs += [b"a" * 70000 + b" b", b" " * 66000 + b"x."]
        x, o = td_corpus.code(1 << 20, seed=3)
        docs += [x[o[d]:o[d + 1]].tobytes() for d in range(0, len(o) - 1, 3)][:150]
        text, offs = H.pack_docs(docs)
        toks, toffs = tok.encode_batch(text, offs)
        _, etoks, eoffs = R.encode_batch(np.frombuffer(text, dtype=np.uint8), offs, n_threads=8, want_tokens=True)
        assert np.array_equal(toffs, eoffs), name
        assert np.array_equal(toks, etoks), name
        assert list(tok.encode_batch(b"snake_case", np.asarray([0, 10], dtype=np.int64))[0]) == list(R.encode(b"snake_case"))
        tok.close()


@pytest.mark.gpu
@pytest.mark.skipif(not ref.available(), reason="compiled reference (oracle/_ref) not built")
def test_gpu_python_surface_with_the_reference_tests_pattern():
    """tests/autogenned_test.py:58-70 of the reference: Encoding(pat_str=r"[a-zA-Z]+|\\s+|[0-9]+|[^\\w\\s]", ...).encode(prompt)."""
    import tokendagger as tiktoken
    _, mr, special = H.llama4()
    enc = tiktoken.Encoding(name="test_tokenizer", pat_str=AUTOGEN, mergeable_ranks=mr, special_tokens=special)
    R = ref.RefTokenizer(AUTOGEN, mr, special)
    for s in ["This is a test prompt for tokenization.", "snake_case_name = 42", "é", "", "a", "naïve café_1", "x" * 5000 + " _ " + "y" * 70]:
        assert enc.encode(s) == list(R.encode(s.encode("utf-8"))), s
        assert enc.decode(enc.encode(s)) == b"".join(R.split_pieces(s.encode("utf-8"))).decode("utf-8"), s  # (skipped text is gone)
    batch = ["one two", "three_four", "5 six!"]
    assert enc.encode_batch(batch) == [list(R.encode(b.encode())) for b in batch]


@pytest.mark.gpu
@pytest.mark.skipif(not ref.available(), reason="compiled reference (oracle/_ref) not built")
def test_gpu_left_context_behind_special_cuts_equals_the_reference():
    """The reference matches a segment behind a special token with the text in front of it as left context (pcre2_match on
    text[0, end) from start_offset, tiktoken.cpp:86-93): \\A and ^ cannot match there, \\b and a one-character look-behind see the
    special's last character.  Round 3 refused such cuts; round 4 sends every segment down with that character in front of it,
    marked as context (EncodeArgs::gx_prefix).  Against CoreBPE::encode(text, {one special}) of the compiled reference (with
    ONE allowed special its segmentation loop is deterministic, tests/test_special_reference.py), batch and single-string forms.
    The patterns here match every character: where a segment's tail matches nothing, the reference's split_text pushes
    text.substr(start_offset) — the rest of the WHOLE text, special token and all, tiktoken.cpp:99 ignores end_offset — as one
    piece and then goes on behind the special token, so that text is tokenized twice; that is not reproduced (the tail piece
    ends where the segment ends)."""
    import tokendagger as tiktoken
    from tokendagger_amd import capi
    _, mr, special = H.llama4()
    rng = random.Random(21)
    names = ["<|begin_of_text|>", "<|eot|>", "<|header_start|>"]
    frags = ["one", " two", "three", "x", " ", "\n", "#", "##", "9", " 42", "naïve", "中文", "_", "-", "a1", "", "s", " s"]
    total = {"wordb": PATTERNS["wordb"] + r"|\s+", "anchors": PATTERNS["anchors"], "lookb1": PATTERNS["lookb1"], "lookb2": PATTERNS["lookb2"] + r"|\s",
             "lookb3": PATTERNS["lookb3"] + r"|\n", "bnd2": PATTERNS["bnd2"] + r"|\s", "words": PATTERNS["words"]}
    for pname, pat in total.items():
        tok = capi.HipTokenizer(pat, mr, special, device=0)
        R = ref.RefTokenizer(pat, mr, special)
        enc = tiktoken.Encoding(name=pname, pat_str=pat, mergeable_ranks=mr, special_tokens=special)
        for name in names:
            docs = []
            for _ in range(120):
                parts = []
                for _ in range(rng.randrange(0, 8)):
                    parts.append(rng.choice(frags))
                    if rng.random() < 0.5:
                        parts.append(rng.choice(names))  # (the allowed one, or another special's literal: ordinary text then)
                docs.append("".join(parts).encode("utf-8"))
            docs += [name.encode(), (name * 2 + "a").encode(), ("a" + name).encode(), (name + "a b" + name + " c").encode(), b"", ("é" + name + "é").encode()]
            blob, offs = H.pack_docs(docs)
            toks, toffs = tok.encode_batch_with_special_strs(blob, offs, [name])
            for i, d in enumerate(docs):
                want = R.encode_special(d, [name]).tolist()
                assert toks[toffs[i]:toffs[i + 1]].tolist() == want, (pname, name, d)
            for d in docs[:25]:  # the single-string form of the Python surface (CoreBPE.encode(text, allowed_special))
                assert enc.encode(d.decode("utf-8"), allowed_special={name}) == R.encode_special(d, [name]).tolist(), (pname, name, d)
        tok.close()


@pytest.mark.gpu
@pytest.mark.skipif(not ref.available(), reason="compiled reference (oracle/_ref) not built")
def test_gpu_generic_chunks_inside_large_documents():
    """The generic engine matches 1 KiB chunks speculatively and checks them against their predecessors (td_generic.hip): one
    document of megabytes, documents that start and end anywhere relative to the chunk grid, multi-byte characters and skipped
    text across chunk boundaries, pieces longer than a chunk (every chunk inside leaves with the same exit), and patterns whose
    matches depend on look-ahead — all against PCRE2 running the very pattern."""
    import td_corpus
    from tokendagger_amd import capi
    _, mr, special = H.llama4()
    rng = random.Random(12)
    eng, _ = td_corpus.english(3 << 20, seed=8)
    mix, _ = td_corpus.mixed(2 << 20, seed=8)
    code, _ = td_corpus.code(1 << 20, seed=8)
    eng, mix, code = eng.tobytes(), mix.tobytes(), code.tobytes()
    snake = ("snake_case_name = naïve_café_%d; " * 40000 % tuple(range(40000))).encode()
    longrun = eng[:3000] + b"a" * 5000 + b" " * 3000 + b"_" * 2500 + eng[:3000] + "é".encode() * 2000 + eng[:5000]
    for name in ("autogen", "words", "look", "cats", "lit", "wordb", "scripts", "lazy1", "lookb1"):
        pat = PATTERNS[name]
        tok = capi.HipTokenizer(pat, mr, special, device=0)
        R = ref.RefTokenizer(pat, mr, special)
        for what, text, offs in (
                ("english, one document", eng, [0, len(eng)]),
                ("mixed-script, one document", mix, [0, len(mix)]),
                ("code + snake_case (skipped text), one document", code + snake, [0, len(code) + len(snake)]),
                ("long runs", longrun, [0, len(longrun)]),
                ("documents cut anywhere", eng[:1 << 20], sorted(set([0, 1 << 20] + [rng.randrange(1 << 20) for _ in range(700)]))),
                ("documents around the chunk grid", eng[:40000], sorted(set([0, 40000, 1023, 1024, 1025, 2047, 2048, 2049, 3072, 5000, 5001, 9216])))):
            if name in ("look", "lit") and what == "long runs":
                continue  # (these back out of a long run one character at a time at every start: quadratic for PCRE2 and for a lane alike)
            offs = np.asarray(offs, dtype=np.int64)
            toks, toffs = tok.encode_batch(text, offs)
            _, etoks, eoffs = R.encode_batch(np.frombuffer(text, dtype=np.uint8), offs, n_threads=8, want_tokens=True)
            assert np.array_equal(toffs, eoffs), (name, what)
            assert np.array_equal(toks, etoks), (name, what)
        tok.close()


@pytest.mark.gpu
@pytest.mark.skipif(not ref.available(), reason="compiled reference (oracle/_ref) not built")
def test_gpu_generic_chunks_that_never_resynchronise():
    """Counted repeats do not resynchronise: in a long run of digits under {1,3} where a chunk's speculative run groups them
    depends on where it started, so most chunks of the run fail the check, and a chunk BEHIND a failed one may pass it
    against that one's speculative (wrong) exit (3-byte fullwidth digits from offset 0 with 64-byte chunks —
    chunk 1 fails, chunks 2 and 3 "validate" against the wrong grouping, chunk 4 fails again with a wrong entry).  One lane per
    document puts it right from the first failed chunk on (td_generic_redo); runs in the middle of ordinary text exercise its
    jump over the chunks that are back in step."""
    import td_corpus
    from tokendagger_amd import capi
    _, mr, special = H.llama4()
    eng, _ = td_corpus.english(1 << 20, seed=18)
    eng = eng.tobytes()
    fw = "１２３４５６７８９０".encode()          # 3 bytes each
    ar = "٠١٢٣٤٥٦٧٨٩".encode()                   # 2 bytes each
    # (every document a few KB: the first three patterns take "everything that is no digit" as ONE piece, and the reference's
    # merge loop is quadratic in the piece length)
    small = [fw * 400,                              # the ADVICE case: from offset 0, nothing but digits
             b"7" * 5000,
             eng[:1500] + b"1234567890" * 700 + eng[5000:6200] + fw * 333 + b" x " + ar * 500 + eng[9000:9900],
             b"ab" + fw * 50 + b"12" + fw * 77 + b"3" + ar * 99 + b".",
             eng[20000:21000] + b"0" * 1025 + eng[300000:301700] + fw * 1000 + eng[600000:600800],
             (b"9" * 70 + b" ") * 300]
    big = small + [eng[20000:300000] + b"0" * 1025 + eng[300000:600000] + fw * 1000 + eng[600000:]]
    for pat, docs in ((r"\p{N}{1,3}|\P{N}+", small), (r"\d{1,3}|\D+", small), (r" ?\p{N}{2,4}|[^\p{N}]+|\p{N}", small),
                      (r"\p{L}+|\p{N}{1,3}| ?[^\s\p{L}\p{N}]+|\s+", big)):
        tok = capi.HipTokenizer(pat, mr, special, device=0)
        R = ref.RefTokenizer(pat, mr, special)
        for cut in (None, 1, 37):
            dd = docs if cut is None else [d[cut:] for d in docs]   # (other alignments against the chunk grid)
            text = b"".join(dd)
            offs = np.concatenate([[0], np.cumsum([len(d) for d in dd])]).astype(np.int64)
            views = [("documents", offs)]
            if cut is None and docs is small:  # (a cut may fall inside a character: as ONE document that is the same text again)
                views.append(("one document", np.asarray([0, len(text)], dtype=np.int64)))
            for view, o in views:
                toks, toffs = tok.encode_batch(text, o)
                _, etoks, eoffs = R.encode_batch(np.frombuffer(text, dtype=np.uint8), o, n_threads=8, want_tokens=True)
                assert np.array_equal(toffs, eoffs), (pat, cut, view)
                assert np.array_equal(toks, etoks), (pat, cut, view)
        tok.close()


@pytest.mark.skipif(not ref.available(), reason="compiled reference (oracle/_ref) not built")
def test_horizontal_space_and_posix_classes_on_their_boundary_characters():
    """\\h is PCRE2's list of 19 characters, [[:alpha:]] ... what PCRE2_UCP turns them into: every listed character and its
    neighbours, the C0 / C1 controls, letters, marks and digits of several scripts, against PCRE2."""
    _, mr, special = H.llama4()
    hs = [0x09, 0x20, 0xA0, 0x1680, 0x180E, 0x2000, 0x2005, 0x200A, 0x200B, 0x202F, 0x205F, 0x3000, 0x3001, 0x1FFF, 0x0A, 0x0B, 0x0C, 0x0D, 0x85, 0x2028, 0x2029]
    chars = [chr(c) for c in hs] + [chr(c) for c in (0x00, 0x1F, 0x7F, 0x80, 0x9F, 0xAD)] + list("aZéßǅʰΩж中あ가٣३௧Ⅷ½_-'!§€〆́⃝")
    rng = random.Random(9)
    docs = ["".join(chars)] + ["".join(rng.choice(chars) for _ in range(rng.randrange(1, 40))) for _ in range(400)]
    for name in ("horiz", "horiz2", "vert", "posix1", "posix2"):
        R = ref.RefTokenizer(PATTERNS[name], mr, special)
        for d in docs:
            b = d.encode("utf-8")
            assert [b[a:e] for a, e in H.rx_split(PATTERNS[name], b)] == R.split_pieces(b), (name, d)


# ---- random patterns of the supported grammar against PCRE2 ---------------------------------------------------------------
_BYTES_ONLY = {bytes([i]): i for i in range(256)}  # (split-only comparisons: the reference is built per pattern, keep its vocabulary tiny)


_FZ_ATOMS = [r"\s", r"\S", r"\d", r"\w", r"\W", r"\p{L}", r"\p{N}", r"\p{Lu}", r"\p{Ll}", r"[a-z]", r"[A-Z0-9_]", r"[^\s\p{L}\p{N}]", r"[^a-c\n]", r".", r"\h",
             r"[[:alpha:]]", "a", "b", " ", r"\n", "x", "é", "中", r"\.", "-", r"\v", r"\V", r"\H", r"\N", r"\p{Han}", r"[\p{Latin}0-9]", r"[[:upper:][:digit:]]",
             r"[^[:space:]]", r"\P{L}", r"\x41", r"\x{e9}", r"[\x{4e00}-\x{9fff}]", r"\D", r"\p{P}", r"\p{Zs}"]
_FZ_QUANT = ["", "", "", "?", "*", "+", "{1,3}", "{2}", "{0,2}", "?+", "*+", "++", "??", "*?", "+?", "{1,3}?", "{2,}+"]
_FZ_GROUPS = [r"(?:ab|a)", r"(?i:the|an|a)", r"(?:x|y|[01])", r"(?:'s|'t)", r"(?:a|b|[xy])", r"(?i:k|s)", r"(?:é|中)"]
_FZ_GQ = ["", "?", "?+", "??", "+", "*", "{1,2}", "++", "*?"]
_FZ_ZW = [r"\A", r"\Z", r"(?<![0-9])", r"(?=\p{L})", r"(?=\s)", r"(?!\S)", r"(?=[0-9])", r"(?!a)", r"\b", r"\B", "^", "$", r"\z", r"(?<=a)", r"(?<!\s)", r"(?<=\p{L})"]


def _random_pattern(rng):
    def alt():
        parts = []
        for _ in range(rng.randrange(1, 4)):
            r = rng.random()
            if r < 0.70:
                parts.append(rng.choice(_FZ_ATOMS) + rng.choice(_FZ_QUANT))
            elif r < 0.85:
                parts.append(rng.choice(_FZ_GROUPS) + rng.choice(_FZ_GQ))
            else:
                parts.append(rng.choice(_FZ_ZW))
        return "".join(parts)
    return "|".join(alt() for _ in range(rng.randrange(1, 5)))


@pytest.mark.skipif(not ref.interp_available(), reason="PCRE2 interpreter oracle (oracle/_ref/libpcre2interp.so) not built")
@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6, 7, 8])
def test_random_patterns_equal_pcre2(seed):
    """Random alternations of quantified classes (greedy, possessive, lazy), literal groups (optional, atomic, lazy) and
    zero-width assertions, each on 130 strings.  This is what found PCRE2's auto-possessification reaching into "(?:..)?+"
    (below).  The oracle is PCRE2's INTERPRETER behind the reference's split loop (oracle/pcre2_interp.c), not the compiled
    reference: the reference JIT-compiles its pattern, and the JIT of the PCRE2 linked here (10.39) has bugs of its own on
    such patterns — it does not find "ab" in "-ab" with (?:ab|a)x*b, nor the two blanks of "K 9  A" with
    \\p{P}??\\p{Zs}+<blank>; the interpreter and this implementation do (oracle/pcre2_probe.c)."""
    rng = random.Random(seed)
    al = " \t\n\r_aAbBxXyY019.,'-éÉ中ſK  "
    strings = ["".join(rng.choice(al) for _ in range(rng.randrange(0, 40))) for _ in range(120)]
    strings += ["", "a", "ab", "the an a", "x01y", "a  b", "aaa", "'s't", "ABC abc 123", "\n\n", " \t "]
    for _ in range(250):
        pat = _random_pattern(rng)
        try:
            H.rx_split(pat, b"abc")
        except ValueError:
            continue  # (outside the subset: rejected, never approximated)
        for s in strings:
            b = s.encode("utf-8")
            assert [b[a:e] for a, e in H.rx_split(pat, b)] == ref.interp_split(pat, b), (pat, s)


@pytest.mark.skipif(not ref.available(), reason="compiled reference (oracle/_ref) not built")
def test_auto_possessification_in_front_of_a_possessive_optional_group():
    """PCRE2 (10.39, the one the reference links here) makes a greedy quantified class possessive when the item behind it
    cannot start with one of its members — and behind the class it looks INTO a possessive optional group and not past it:
    'a+(?:q)?+a' never gives an 'a' back and does not match 'aaa'.  The reference is what PCRE2 does; td_regex.cpp marks such
    classes possessive (only those: a group that can start with a member, a lazy class, an assertion in between are left alone)."""
    _, mr, special = H.llama4()
    cases = [(r"a+(?:q)?+a", "xaaa1"), (r"a+(?:a)?+a", "xaaa1"), (r"a?(?:q)?+a", "xa1"), (r"a*(?:q)?+a", "xaa1"), (r"[ab]+(?:q)?+a", "xaba1"),
             (r"[ab]+(?:b)?+a", "xaba1"), (r"a+(?:q|a)?+a", "xaaa1"), (r"a+(?:q)?+[ab]", "xaaa1"), (r"a+(?:q)?+\w", "xaaa1"), (r"\w+(?:q)?+a", "xaaa1"),
             (r"\w+(?:')?+a", "xaaa1"), (r"a+(?:q)?+(?:q)?+a", "xaaa1"), (r"a+(?:q)?+b?a", "xaaa1"), (r"a+(?=a)(?:q)?+a", "xaaa1"),
             (r"a{1,3}(?:q)?+a", "xaaa1"), (r"a+?(?:q)?+a1", "xaaa1"), (r"(?:b)?(?:q)?+a", "xba1"), (r"s+(?i:K)?+s", "xsss1"), (r"[k\x{212A}]+(?i:K)?+k", "xkkk1"),
             (r"(?:x|y|[01])+(?i:the|an|a)?+\D", ",0x"), (r"[xy01]+(?i:the|an|a)?+\D", ",0x"), (r" +(?:'s|'t)?+ {2,}+", "x   1"), (r"\d{0,2}(?i:the|an|a)?+\d{2}", " 901."), (r"\p{L}+(?:'s|'t)?+\S+?", "Kbé\ta")]
    for pat, s in cases:
        b = s.encode("utf-8")
        R = ref.RefTokenizer(pat, mr, special)
        assert [b[a:e] for a, e in H.rx_split(pat, b)] == R.split_pieces(b), (pat, s)
    # the decision class by class: 31 quantified classes x 7 groups x 4 quantifiers, each on a subject that matches only if the
    # class gives a character back.  What the table of PCRE2's answers looks like: possessive exactly when no alternative of
    # the group can start with a member of the class — except that a group with a bracket of several characters among its
    # alternatives is compared only with classes written without class escapes, properties and POSIX names.
    bases = [("a", "a"), ("[ab]", "a"), ("[^xy]", "a"), ("[aé]", "a"), (r"[a\p{Lu}]", "a"), (r"\d", "1"), (r"\w", "a"), (r"\s", " "), (r"\p{L}", "a"), (r"\P{N}", "a"),
             (".", "a"), ("[[:upper:]]", "A"), (r"[^\s\p{L}]", "1"), (r"\S", "a"), (r"\D", "a"), (r"\h", " "), (r"\p{Han}", "中"), (r"[\p{Han}]", "中"), (r"\N", "a"),
             ("[a-z]", "a"), (r"\W", "-"), ("[^qrs]", "a"), ("[^q]", "a"), ("[a-c]", "a"), ("[a-cé-ü]", "a"), (r"[^\d]", "a"), (r"[^a-c\n]", "x"), (r"[ab\d]", "a"),
             (r"[\x{4e00}-\x{9fff}]", "中"), ("b", "b"), ("é", "é")]
    groups = ["(?:q)?+", "(?:q|[rs])?+", "(?i:q)?+", "(?:ü)?+", "(?:[rs])?+", "(?:%|#)?+", "(?:%|[#])?+"]
    n = 0
    for base, ch in bases:
        for g in groups:
            for q in ("+", "*", "{1,2}", "?"):
                pat = base + q + g + (r"\x20" if ch == " " else ch)
                b = (("x" + ch * 2) if q != "?" else ("-" + ch)).encode("utf-8") + b"1"
                R = ref.RefTokenizer(pat, _BYTES_ONLY, {})
                assert [b[a:e] for a, e in H.rx_split(pat, b)] == R.split_pieces(b), (pat, b)
                n += 1
    assert n == 31 * 7 * 4
