"""Seeded synthetic corpora for bench.py and the tests (data only; no tokenization here).

`english(n_bytes, seed)` reproduces the SHAPE of the reference's throughput corpus
(/root/reference/tests/throughput_test.py:246-333, `generate_test_text`): paragraphs of 50-200 words
drawn uniformly from its common-word list, joined by single spaces, first letter capitalised, 2-5
attempts to drop one of `, . ! ?` in front of a space at a random character position in [10, len-10],
each paragraph terminated by a blank line, the whole trimmed to exactly n_bytes.  The reference draws
from Python's unseeded `random`; here the draws come from a seeded numpy Generator and are
vectorised (a 1 GiB corpus takes seconds, not minutes), so CPU baseline and GPU consume the same
bytes.  Documents are the paragraphs (returned as int64 offsets).

`mixed(n_bytes, seed)`  : new in this repo (the reference has no mixed-language generator, SURVEY 8d
config 4): Latin / CJK / Cyrillic / Arabic / Devanagari / emoji words, digits and punctuation.
`code(n_bytes, seed)`   : source-code-like lines (identifiers, operators, indentation, long runs).
"""
from __future__ import annotations

import numpy as np

# word list of the reference generator (throughput_test.py:251-281), duplicates kept: they weight the draw
_WORDS = (
    "the be to of and a in that have i it for not on with he as you do at this but his by from they we say her she "
    "or an will my one all would there their what so up out if about who get which go me when make can like time no "
    "just him know take people into year your good some could them see other than then now look only come its over "
    "think also back after use two how our work first well way even new want because any these give day most us is "
    "was are been has had were said each which their time will about if up out many then them these so some her "
    "would make like into him you could more go no way could my than first water been call who its now find long "
    "down day did get come made may part over new sound take only little work know place year live me back give "
    "most very after thing our just name good sentence man think say great where help through much before line "
    "right too mean old any same tell boy follow came want show also around form three small set put end why again "
    "turn here off went old number great tell men say small every found still between mane should home big give "
    "air line set own under read last never us left end along while might next sound below saw something thought "
    "both few those always looked show large often together asked house don't world going want school important "
    "until form food keep children feet land side without boy once animal life enough took four"
).split()


def _word_table(words):
    enc = [w.encode("utf-8") for w in words]
    lens = np.asarray([len(w) for w in enc], dtype=np.int64)
    offs = np.zeros(len(enc) + 1, dtype=np.int64)
    np.cumsum(lens, out=offs[1:])
    blob = np.frombuffer(b"".join(enc), dtype=np.uint8)
    return blob, offs, lens


def _assemble(blob, offs, lens, ids, sep_after):
    """Concatenate words `ids`, each followed by the byte string selected by sep_after (0: nothing)."""
    wl = lens[ids]
    starts = np.zeros(len(ids) + 1, dtype=np.int64)
    np.cumsum(wl + sep_after, out=starts[1:])
    total = int(starts[-1])
    out = np.empty(total, dtype=np.uint8)
    # position -> (word index, offset inside word)
    widx = np.repeat(np.arange(len(ids)), wl + sep_after)
    inner = np.arange(total) - starts[:-1][widx]
    in_word = inner < wl[widx]
    src = offs[:-1][ids][widx] + np.minimum(inner, wl[widx] - 1)
    out[:] = blob[src]
    return out, starts, in_word, widx, inner


def english(n_bytes: int, seed: int = 0, block_bytes: int = 1 << 24):
    """-> (uint8[n_bytes], int64 doc_offsets) ; documents = paragraphs."""
    rng = np.random.default_rng(seed)
    blob, offs, lens = _word_table(_WORDS)
    mean_word = float(lens.mean()) + 1.0
    parts, doc_lens = [], []
    have = 0
    while have < n_bytes:
        want = min(block_bytes, n_bytes - have + 4096)
        n_par = max(1, int(want / (125 * mean_word)) + 1)
        par_words = rng.integers(50, 201, size=n_par)
        n_words = int(par_words.sum())
        ids = rng.integers(0, len(_WORDS), size=n_words)
        last_of_par = np.zeros(n_words, dtype=bool)
        last_of_par[np.cumsum(par_words) - 1] = True
        sep = np.where(last_of_par, 2, 1).astype(np.int64)  # " " between words, "\n\n" after a paragraph
        out, starts, in_word, widx, inner = _assemble(blob, offs, lens, ids, sep)
        sep_pos = ~in_word
        out[sep_pos] = np.where(last_of_par[widx[sep_pos]], ord("\n"), ord(" "))
        # capitalise the first letter of each paragraph
        first_word = np.concatenate([[0], np.cumsum(par_words)[:-1]])
        fpos = starts[:-1][first_word]
        c = out[fpos]
        out[fpos] = np.where((c >= 97) & (c <= 122), c - 32, c)
        # punctuation: 2-5 attempts per paragraph at a uniform char position in [10, len-10]; it lands only on a space
        par_start = starts[:-1][first_word]
        par_end = np.concatenate([par_start[1:], [len(out)]]) - 2
        attempts = rng.integers(2, 6, size=n_par)
        pidx = np.repeat(np.arange(n_par), attempts)
        lo = par_start[pidx] + 10
        hi = np.maximum(par_end[pidx] - 10, lo)
        pos = lo + (rng.random(len(pidx)) * (hi - lo + 1)).astype(np.int64)
        pos = np.unique(pos[out[pos] == ord(" ")])
        marks = np.frombuffer(b",.!?", dtype=np.uint8)[rng.integers(0, 4, size=len(pos))]
        out = np.insert(out, pos, marks)
        # paragraph (document) lengths after insertion
        ins_before = np.searchsorted(pos, par_start, side="left")
        new_start = par_start + ins_before
        new_end = np.concatenate([new_start[1:], [len(out)]])
        parts.append(out)
        doc_lens.append(new_end - new_start)
        have += len(out)
    text = np.concatenate(parts)[:n_bytes]
    dl = np.concatenate(doc_lens)
    doc_offs = np.zeros(len(dl) + 1, dtype=np.int64)
    np.cumsum(dl, out=doc_offs[1:])
    keep = int(np.searchsorted(doc_offs, n_bytes, side="left"))
    doc_offs = doc_offs[:keep + 1].copy()
    doc_offs[-1] = n_bytes
    if len(doc_offs) >= 2 and doc_offs[-2] >= n_bytes:
        doc_offs = doc_offs[:-1]
        doc_offs[-1] = n_bytes
    return np.ascontiguousarray(text), doc_offs


def chunk_offsets(n_bytes: int, n_chunks: int) -> np.ndarray:
    """The reference benchmark's document boundaries: n_chunks equal slices (throughput_test.py:399-410).
    Only valid for ASCII corpora (slices are by character there)."""
    size = n_bytes // n_chunks
    offs = np.arange(n_chunks + 1, dtype=np.int64) * size
    offs[-1] = n_bytes
    return offs


_MIXED = {
    "latin": "the of and to in is you that it he was for on are as with his they I at be this have from or one had by "
             "word but not what all were we when your can said there use an each which she do how their if will up "
             "Über straße café naïve résumé señor garçon Ångström".split(),
    "cjk": "的 一 是 不 了 人 我 在 有 他 这 为 之 大 来 以 个 中 上 们 到 说 国 和 地 也 子 时 道 出 而 要 于 就 下 得 可 你 年 生 "
           "東京 日本語 こんにちは ありがとう 世界 カタカナ 한국어 안녕하세요 감사합니다".split(),
    "cyrillic": "и в не на я быть он с что а по это она этот к но они мы как из у который то за свой что весь год от так о "
                "Москва Россия привет спасибо".split(),
    "arabic": "في من على أن إلى هذا كان ما لا هو التي الذي عن مع هذه كل بعد قد بين ذلك حيث كما عند لم غير "
              "مُحَمَّد السَّلَامُ".split(),
    "devanagari": "के है में की और से का को पर यह कि एक हैं भी नहीं तो ही या था हो इस कर लिए अपने ने साथ "
                  "नमस्ते धन्यवाद हिन्दी".split(),
    "emoji": ["\U0001F600", "\U0001F680", "✨", "\U0001F468‍\U0001F4BB", "\U0001F1FA\U0001F1F8",
              "\U0001F44D\U0001F3FD", "❤️", "\U0001F525", "\U0001F3F3️‍\U0001F308", "\U0001F389"],
}


def mixed(n_bytes: int, seed: int = 0, block_words: int = 1 << 20):
    """Mixed-script corpus; documents are lines. -> (uint8[n_bytes], int64 doc_offsets)"""
    rng = np.random.default_rng(seed)
    scripts = list(_MIXED)
    weights = np.asarray([0.42, 0.18, 0.14, 0.10, 0.10, 0.06])
    tables = {k: _word_table(v) for k, v in _MIXED.items()}
    parts = []
    have = 0
    while have < n_bytes + 64:
        chunks = []
        # sentences of one script each
        n_sent = max(8, block_words // 12)
        which = rng.choice(len(scripts), size=n_sent, p=weights)
        for si in range(n_sent):
            blob, offs, lens = tables[scripts[which[si]]]
            nw = int(rng.integers(3, 20))
            ids = rng.integers(0, len(lens), size=nw)
            joiner = b"" if scripts[which[si]] == "cjk" and rng.random() < 0.7 else b" "
            words = [bytes(blob[offs[i]:offs[i + 1]]) for i in ids]
            if rng.random() < 0.25:
                words.insert(int(rng.integers(0, len(words) + 1)), str(int(rng.integers(0, 10 ** int(rng.integers(1, 8))))).encode())
            sent = joiner.join(words)
            end = [b". ", b"! ", b"? ", "。".encode(), b", ", b"\n", b"\n\n", b": ", b" - "][int(rng.integers(0, 9))]
            chunks.append(sent + end)
            have += len(sent) + len(end)
            if have >= n_bytes + 64:
                break
        parts.append(b"".join(chunks))
    raw = b"".join(parts)
    # cut on a character boundary at or below n_bytes, pad with spaces to the exact size
    cut = n_bytes
    while cut > 0 and (raw[cut] & 0xC0) == 0x80:
        cut -= 1
    text = np.frombuffer(raw[:cut] + b" " * (n_bytes - cut), dtype=np.uint8).copy()
    nl = np.nonzero(text == 10)[0] + 1
    doc_offs = np.unique(np.concatenate([[0], nl[nl < n_bytes], [n_bytes]])).astype(np.int64)
    return text, doc_offs


_CODE_LINES = [
    "def {id}({id}, {id}=None):", "    return {id}.{id}({num}) + {id}[{num}]", "    if {id} is not None and {id} > {num}:",
    "for (int {id} = 0; {id} < {id}.size(); ++{id}) {{", "    std::vector<std::pair<size_t, int>> {id};", "}}",
    "// {id} {id} {id} {id}", "# TODO: {id} the {id} before {id}", "    {id} = {{\"{id}\": {num}, \"{id}\": [{num}, {num}]}}",
    "#include <{id}/{id}.h>", "import {id}.{id} as {id}", "    printf(\"%d %s\\n\", {id}, {id}->{id});",
    "const {id} = async ({id}) => {{ await {id}.{id}(); }};", "\t\t{id} += {id} * {num};", "",
    "    x = 0x{num}ULL << {num};  // {id}", "template <class {id}> struct {id} : public {id}<{id}> {{",
    "    self.{id}_{id}_{id} = {id}_{id}  # {id}", "        \"{id}\": \"{id} {id} {id}\",",
]
_CODE_RULER = "/* ==================================================================== */"
_IDS = ("i j k n x y tmp value result index count data buffer node left right parent key item list map size length "
        "offset start end pos token tokens text encode decode rank piece merge lookup table hash vocab special regex "
        "tokenizer_config mergeable_ranks get_thread_local_match_data find_next_special_token byte_pair_encode").split()


def code(n_bytes: int, seed: int = 0):
    """Source-code-like corpus; documents are 'files' of 20-400 lines."""
    rng = np.random.default_rng(seed)
    out, doc_offs, have = [], [0], 0
    while have < n_bytes:
        nlines = int(rng.integers(20, 400))
        lines = []
        for _ in range(nlines):
            tpl = _CODE_LINES[int(rng.integers(0, len(_CODE_LINES)))]
            if rng.random() < 1.0 / 800:  # comment rulers: the pieces longer than 64 bytes of real source files
                tpl = _CODE_RULER       # (99.74 % of the bytes are in pieces <= 64 B)
            n_id, n_num = tpl.count("{id}"), tpl.count("{num}")
            ids = [_IDS[int(i)] for i in rng.integers(0, len(_IDS), size=n_id)]
            nums = [str(int(rng.integers(0, 10 ** int(rng.integers(1, 7))))) for _ in range(n_num)]
            it_id, it_num = iter(ids), iter(nums)
            s = tpl.replace("{{", "\x01").replace("}}", "\x02")
            while "{id}" in s:
                s = s.replace("{id}", next(it_id), 1)
            while "{num}" in s:
                s = s.replace("{num}", next(it_num), 1)
            lines.append(s.replace("\x01", "{").replace("\x02", "}"))
        blob = ("\n".join(lines) + "\n").encode("utf-8")
        out.append(blob)
        have += len(blob)
        doc_offs.append(min(have, n_bytes))
    text = np.frombuffer(b"".join(out)[:n_bytes], dtype=np.uint8).copy()
    doc_offs = np.unique(np.asarray(doc_offs, dtype=np.int64))
    doc_offs[-1] = n_bytes
    return text, doc_offs


def code_files(n_bytes: int = 0):
    """BASELINE config 5's real input: the file set of the reference's tests/code_performance_benchmark.py (21 files,
    2 146 667 bytes, one document per file; fixture tests/golden/code_corpus.npz made by tools/make_code_corpus.py),
    repeated WHOLE floor(n_bytes / set size) times (at least once) — the result is a multiple of the set, not n_bytes.
    -> (uint8[reps * set], int64 doc_offsets)"""
    from pathlib import Path
    g = np.load(Path(__file__).resolve().parent / "tests" / "golden" / "code_corpus.npz", allow_pickle=False)
    unit, uo = g["text"], g["offsets"].astype(np.int64)
    reps = max(1, n_bytes // len(unit))
    x = np.tile(unit, reps)
    offs = np.concatenate([uo[:-1] + r * len(unit) for r in range(reps)] + [[reps * len(unit)]]).astype(np.int64)
    return np.ascontiguousarray(x), offs


def chat(n_bytes: int, seed: int = 0):
    """Chat-formatted corpus: the English generator's paragraphs as turns of conversations in the Llama-4 chat template
    (<|begin_of_text|><|header_start|>role<|header_end|>\n\n ... <|eot|>), one conversation per document.  Special tokens
    are ~3 % of the bytes.  Some turns mention a special token's literal inside running text and a few literals are cut short
    ("<|eot", "<|header_start|"), so the search has near misses to reject.  -> (uint8[n_bytes], int64 doc_offsets)"""
    rng = np.random.default_rng(seed + 77)
    text, offs = english(n_bytes, seed=seed)
    paras = [text[offs[i]:offs[i + 1]].tobytes().rstrip(b"\n") for i in range(len(offs) - 1)]
    out, docs, total, k = [], [0], 0, 0
    roles = [b"system", b"user", b"assistant", b"ipython"]
    near = [b"<|eot", b"<|header_start|", b"<|", b"<|end_of_text|", b"|>", b"<|python_start|>", b"<|image|>", b"<|fim_middle|>"]
    while total < n_bytes and k < len(paras):
        conv = [b"<|begin_of_text|>"]
        for t in range(int(rng.integers(1, 7))):
            if k >= len(paras):
                break
            body = paras[k]
            k += 1
            if rng.random() < 0.08:
                cut = int(rng.integers(0, len(body) + 1))
                body = body[:cut] + near[int(rng.integers(0, len(near)))] + body[cut:]
            conv += [b"<|header_start|>", roles[min(t, 1) if t < 2 else int(rng.integers(1, 4))], b"<|header_end|>\n\n", body,
                     b"<|eot|>" if rng.random() < 0.9 else b"<|eom|>"]
        if rng.random() < 0.3:
            conv.append(b"<|end_of_text|>")
        doc = b"".join(conv)
        out.append(doc)
        total += len(doc)
        docs.append(total)
    buf = np.frombuffer(b"".join(out), dtype=np.uint8)
    if len(buf) >= n_bytes:
        buf = buf[:n_bytes].copy()
        d = np.asarray(docs, dtype=np.int64)
        d = d[d < n_bytes]
        return buf, np.concatenate([d, [n_bytes]]).astype(np.int64)
    return buf.copy(), np.asarray(docs, dtype=np.int64)
