"""Shared test helpers: vocab, oracle handles, CPU twin loader, input generators."""
from __future__ import annotations

import ctypes
import functools
import random
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

from tokendagger_amd import vocab_io  # noqa: E402


@functools.lru_cache(maxsize=None)
def llama4():
    """-> (pat_str, mergeable_ranks incl. special strings (as the reference's tests build it), special_tokens)"""
    name, pat, ranks, special = vocab_io.load_tdv(vocab_io.default_vocab_path())
    mr = dict(ranks)
    for k, v in special.items():  # reference tests add specials into mergeable_ranks (test_tokendagger_vs_tiktoken.py:181-183)
        mr[k.encode("utf-8")] = v
    return pat, mr, special


@functools.lru_cache(maxsize=None)
def ref_tokenizer():
    from oracle import ref
    if not ref.available():
        subprocess.check_call([str(ROOT / "oracle" / "build_ref.sh")])
    pat, mr, special = llama4()
    return ref.RefTokenizer(pat, mr, special)


@functools.lru_cache(maxsize=None)
def port_tokenizer():
    from oracle import port
    if not port.available():
        subprocess.check_call([str(ROOT / "oracle" / "build_oracle.sh")])
    _, mr, _ = llama4()
    return port.OracleTokenizer(mr)


# "tekken-style" configuration: the Mistral tekken split pattern over the Llama-4 vocabulary.  The real tekken.json
# vocabulary is absent from the reference checkout (.MISSING_LARGE_BLOBS), so the pattern family is pinned
# on this labelled surrogate.
TEKKEN_PAT = vocab_io.TEKKEN_PAT_STR


@functools.lru_cache(maxsize=None)
def ref_tokenizer_tekken():
    from oracle import ref
    if not ref.available():
        subprocess.check_call([str(ROOT / "oracle" / "build_ref.sh")])
    _, mr, special = llama4()
    return ref.RefTokenizer(TEKKEN_PAT, mr, special)


@functools.lru_cache(maxsize=None)
def port_tokenizer_tekken():
    from oracle import port
    if not port.available():
        subprocess.check_call([str(ROOT / "oracle" / "build_oracle.sh")])
    _, mr, _ = llama4()
    return port.OracleTokenizer(mr, port.VARIANT_TEKKEN)


CL100K_PAT = vocab_io.CL100K_PAT_STR


@functools.lru_cache(maxsize=None)
def ref_tokenizer_cl100k():
    from oracle import ref
    if not ref.available():
        subprocess.check_call([str(ROOT / "oracle" / "build_ref.sh")])
    _, mr, special = llama4()
    return ref.RefTokenizer(CL100K_PAT, mr, special)


@functools.lru_cache(maxsize=None)
def port_tokenizer_cl100k():
    from oracle import port
    if not port.available():
        subprocess.check_call([str(ROOT / "oracle" / "build_oracle.sh")])
    _, mr, _ = llama4()
    return port.OracleTokenizer(mr, port.VARIANT_CL100K)


GPT2_PAT = vocab_io.GPT2_PAT_STR


@functools.lru_cache(maxsize=None)
def ref_tokenizer_gpt2():
    from oracle import ref
    if not ref.available():
        subprocess.check_call([str(ROOT / "oracle" / "build_ref.sh")])
    _, mr, special = llama4()
    return ref.RefTokenizer(GPT2_PAT, mr, special)


@functools.lru_cache(maxsize=None)
def port_tokenizer_gpt2():
    from oracle import port
    if not port.available():
        subprocess.check_call([str(ROOT / "oracle" / "build_oracle.sh")])
    _, mr, _ = llama4()
    return port.OracleTokenizer(mr, port.VARIANT_GPT2)


def pack_docs(docs: list[bytes]):
    offs = np.zeros(len(docs) + 1, dtype=np.int64)
    np.cumsum([len(d) for d in docs], out=offs[1:])
    return b"".join(docs), offs


def pack_vocab(mergeable_ranks: dict[bytes, int]):
    items = list(mergeable_ranks.items())
    ranks = np.asarray([r for _, r in items], dtype=np.int32)
    offs = np.zeros(len(items) + 1, dtype=np.int64)
    np.cumsum([len(b) for b, _ in items], out=offs[1:])
    blob = np.frombuffer(b"".join(b for b, _ in items) or b"\0", dtype=np.uint8).copy()
    return blob, offs, ranks


# ----------------------------------------------------------------------------- CPU twin -----
TWIN_SO = ROOT / "tests" / "twin" / "_build" / "libtdtwin.so"


def build_twin():
    TWIN_SO.parent.mkdir(parents=True, exist_ok=True)
    srcs = [ROOT / "tests/twin/td_twin.cpp", ROOT / "tokendagger_amd/csrc/td_tables.cpp", ROOT / "tokendagger_amd/csrc/td_regex.cpp"]
    deps = srcs + [ROOT / "tokendagger_amd/csrc/td_common.h", ROOT / "tokendagger_amd/csrc/td_tables.h",
                   ROOT / "tokendagger_amd/csrc/td_regex.h", ROOT / "tokendagger_amd/csrc/generated/unicode_classes.inc",
                   ROOT / "tokendagger_amd/csrc/generated/unicode_gc.inc", ROOT / "tokendagger_amd/csrc/generated/unicode_scripts.inc"]
    if TWIN_SO.exists() and all(TWIN_SO.stat().st_mtime >= d.stat().st_mtime for d in deps):
        return
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
                           *map(str, srcs), "-o", str(TWIN_SO)])


def rx_split(pattern: str, data: bytes):
    """Generic split pattern through td_regex.cpp (compile) + td_regex.h (the matcher the device runs), one document.
    -> list of (start, end), or raises ValueError with the compiler's message when the pattern is not supported."""
    build_twin()
    lib = ctypes.CDLL(str(TWIN_SO))
    lib.twin_rx_split.restype = ctypes.c_int64
    lib.twin_rx_split.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                  ctypes.c_int64, ctypes.c_char_p, ctypes.c_int]
    cap = len(data) + 1
    st = np.empty(cap, dtype=np.int64); en = np.empty(cap, dtype=np.int64)
    err = ctypes.create_string_buffer(512)
    n = lib.twin_rx_split(pattern.encode("utf-8"), data, len(data), st.ctypes.data, en.ctypes.data, cap, err, 512)
    if n == -1:
        raise ValueError(err.value.decode())
    assert n >= 0, n
    return list(zip(st[:n].tolist(), en[:n].tolist()))


class Twin:
    """CPU twin of the device algorithm (tests/twin/td_twin.cpp)."""

    def seeded_merge(self, pieces: list[bytes]):
        """-> (list of id arrays, characters entered whole, parts at the start): the pieces merged from their seeded parts
        (td_common.h: character seeds), as td_long_pieces sets them up."""
        blob, offs = pack_docs(pieces)
        cap = len(blob) + 1
        ids = np.empty(cap, dtype=np.int32)
        io = np.empty(len(pieces) + 1, dtype=np.int64)
        st = np.zeros(2, dtype=np.int64)
        n = self._lib.twin_seeded_merge(self._h, blob, offs.ctypes.data, len(pieces), ids.ctypes.data, cap, io.ctypes.data, st.ctypes.data)
        if n < 0:
            raise RuntimeError(f"twin_seeded_merge: bookkeeping check {n} failed")
        return [ids[io[i]:io[i + 1]] for i in range(len(pieces))], int(st[0]), int(st[1])

    def __init__(self, pat_str: str, mergeable_ranks: dict[bytes, int], special: dict[str, int] | None = None):
        build_twin()
        lib = ctypes.CDLL(str(TWIN_SO))
        lib.twin_create.restype = ctypes.c_void_p
        lib.twin_create.argtypes = [ctypes.c_char_p, ctypes.c_int64] + [ctypes.c_void_p] * 3 + [ctypes.c_int64] + \
            [ctypes.c_void_p] * 3 + [ctypes.POINTER(ctypes.c_int)]
        lib.twin_destroy.argtypes = [ctypes.c_void_p]
        lib.twin_info.restype = ctypes.c_int64
        lib.twin_info.argtypes = [ctypes.c_void_p, ctypes.c_int]
        lib.twin_classify.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int64, ctypes.c_void_p,
                                      ctypes.c_int64, ctypes.c_void_p]
        lib.twin_split_serial.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int64, ctypes.c_void_p,
                                          ctypes.c_int64, ctypes.c_void_p]
        lib.twin_split_tiled.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int64, ctypes.c_void_p,
                                         ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        lib.twin_sync_violations.restype = ctypes.c_int64
        lib.twin_sync_violations.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int64, ctypes.c_void_p,
                                             ctypes.c_int64, ctypes.c_void_p]
        lib.twin_bits_check.restype = ctypes.c_int64
        lib.twin_bits_check.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int64, ctypes.c_void_p,
                                        ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
        lib.twin_arrmask_check.restype = ctypes.c_int64
        lib.twin_arrmask_check.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int64, ctypes.c_void_p,
                                           ctypes.c_int64, ctypes.c_void_p]
        lib.twin_encode.restype = ctypes.c_int64
        lib.twin_encode.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int64, ctypes.c_void_p,
                                    ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
        lib.twin_seeded_merge.restype = ctypes.c_int64
        lib.twin_seeded_merge.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                          ctypes.c_void_p, ctypes.c_void_p]
        self._lib = lib
        special = special or {}
        b, o, r = pack_vocab(mergeable_ranks)
        sb, so, sr = pack_vocab({k.encode("utf-8"): v for k, v in special.items()})
        rc = ctypes.c_int(0)
        self._h = lib.twin_create(pat_str.encode("utf-8"), len(r), b.ctypes.data, o.ctypes.data, r.ctypes.data,
                                  len(sr), sb.ctypes.data, so.ctypes.data, sr.ctypes.data, ctypes.byref(rc))
        self.rc = rc.value
        if not self._h:
            raise RuntimeError(f"twin_create failed rc={rc.value}")

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.twin_destroy(self._h)
            self._h = None

    def info(self, what: int) -> int:
        return self._lib.twin_info(self._h, what)

    @staticmethod
    def _offs(data, offs):
        if offs is None:
            offs = np.asarray([0, len(data)], dtype=np.int64)
        return np.ascontiguousarray(offs, dtype=np.int64)

    def classify(self, data: bytes, offs=None) -> np.ndarray:
        offs = self._offs(data, offs)
        out = np.zeros(max(len(data), 1), dtype=np.uint8)
        self._lib.twin_classify(self._h, data, len(data), offs.ctypes.data, len(offs) - 1, out.ctypes.data)
        return out[:len(data)]

    def split_serial(self, data: bytes, offs=None) -> np.ndarray:
        offs = self._offs(data, offs)
        out = np.zeros(max(len(data), 1), dtype=np.uint8)
        rc = self._lib.twin_split_serial(self._h, data, len(data), offs.ctypes.data, len(offs) - 1, out.ctypes.data)
        assert rc == 0, rc
        return np.nonzero(out[:len(data)])[0]

    def split_tiled(self, data: bytes, offs=None):
        offs = self._offs(data, offs)
        out = np.zeros(max(len(data), 1), dtype=np.uint8)
        stats = np.zeros(8, dtype=np.int64)
        rc = self._lib.twin_split_tiled(self._h, data, len(data), offs.ctypes.data, len(offs) - 1, out.ctypes.data,
                                        None, stats.ctypes.data)
        assert rc == 0, rc
        return np.nonzero(out[:len(data)])[0], stats

    def sync_violations(self, data: bytes, offs=None) -> tuple[int, int]:
        offs = self._offs(data, offs)
        ns = ctypes.c_int64(0)
        bad = self._lib.twin_sync_violations(self._h, data, len(data), offs.ctypes.data, len(offs) - 1, ctypes.byref(ns))
        return int(bad), ns.value

    def bits_check(self, data: bytes, offs=None) -> tuple[int, int, int]:
        """-> (mismatches, unresolved, checked) of the bit-parallel scanner against the byte scanner"""
        offs = self._offs(data, offs)
        un = ctypes.c_int64(0); ck = ctypes.c_int64(0)
        bad = self._lib.twin_bits_check(self._h, data, len(data), offs.ctypes.data, len(offs) - 1, ctypes.byref(un),
                                        ctypes.byref(ck))
        return int(bad), un.value, ck.value

    def fast_stats(self) -> tuple[int, int]:
        """-> (pieces seen, pieces the branch-free scanner answered) since the last call"""
        t = ctypes.c_int64(0); h = ctypes.c_int64(0)
        self._lib.twin_fast_stats(ctypes.byref(t), ctypes.byref(h))
        return t.value, h.value

    def word_rules_check(self, data: bytes, offs=None, head_state=None) -> tuple[int, list[int]]:
        """-> (mismatches, [heads, unresolved heads, pieces, pieces in unresolved regions]) of the whole-word boundary
        rules (split_unresolved_heads) against the byte scanner"""
        offs = self._offs(data, offs)
        st = (ctypes.c_int64 * 4)()
        self._lib.twin_word_rules_check.restype = ctypes.c_int64
        self._lib.twin_word_rules_check.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int64, ctypes.c_void_p,
                                                    ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
        bad = self._lib.twin_word_rules_check(self._h, data, len(data), offs.ctypes.data, len(offs) - 1, st,
                                              None if head_state is None else head_state.ctypes.data)
        return int(bad), list(st)

    def arrmask_check(self, data: bytes, offs=None) -> tuple[int, int]:
        """-> (mismatches, checked) of the array-mask scanner (no run-length limit) against the byte scanner"""
        offs = self._offs(data, offs)
        ck = ctypes.c_int64(0)
        bad = self._lib.twin_arrmask_check(self._h, data, len(data), offs.ctypes.data, len(offs) - 1, ctypes.byref(ck))
        return int(bad), ck.value

    def pair_probe_check(self, n_random: int = 200000, seed: int = 1) -> tuple[int, list[int]]:
        """-> (mismatches, [pairs, second-seat probes for them, other pairs tried, second-seat probes for those, bad byte_pair_id
        entries]) of the pair table probed in the device's order (first seat, the second one on demand: PAIR_FINAL) against both
        seats"""
        st = (ctypes.c_int64 * 5)()
        self._lib.twin_pair_probe_check.restype = ctypes.c_int64
        self._lib.twin_pair_probe_check.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64, ctypes.c_void_p]
        bad = self._lib.twin_pair_probe_check(self._h, n_random, seed, st)
        return int(bad), list(st)

    def encode_batch(self, data: bytes, offs=None, mode: int = 0):
        offs = self._offs(data, offs)
        out = np.empty(max(len(data), 1), dtype=np.int32)
        oo = np.zeros(len(offs), dtype=np.int64)
        n = self._lib.twin_encode(self._h, data, len(data), offs.ctypes.data, len(offs) - 1, mode, out.ctypes.data,
                                  out.size, oo.ctypes.data)
        if n < 0:
            raise RuntimeError(f"twin_encode error {-n}")
        return out[:n].copy(), oo


@functools.lru_cache(maxsize=None)
def twin_llama4():
    pat, mr, special = llama4()
    return Twin(pat, mr, special)


@functools.lru_cache(maxsize=None)
def twin_tekken():
    _, mr, special = llama4()
    return Twin(TEKKEN_PAT, mr, special)


@functools.lru_cache(maxsize=None)
def twin_cl100k():
    _, mr, special = llama4()
    return Twin(CL100K_PAT, mr, special)


@functools.lru_cache(maxsize=None)
def twin_gpt2():
    _, mr, special = llama4()
    return Twin(GPT2_PAT, mr, special)


# ----------------------------------------------------------------------------- inputs -------
# Building blocks for adversarial strings (all written as escapes so this file stays plain ASCII).
FUZZ_ALPHABET = [
    "a", "b", "z", "Z", "Q", "A", "the", "Hello", "WORLD", " ", "  ", "   ", "\n", "\r", "\r\n", "\n\n", "\t", "\x0b", "\x0c",
    "'", "'s", "'S", "'t", "'re", "'RE", "'ve", "'m", "'ll", "'LL", "'d", "'ſ", "'x", "''", "/", "//", "!", "?", ".", ",", ";",
    ":", "-", "_", "=", "(", ")", "{", "}", "[", "]", "<", ">", "|", "\\", "\"", "#", "$", "%", "&", "*", "+", "@", "^", "~", "`",
    "0", "1", "23", "456", "7890", "٣", "५", "Ⅳ", "½", "²",
    "中", "文", "日本語", "한국어", "é", "É", "ñ", "ü", "ß", "İ", "ı",
    "ǅ", "ǈ", "ʰ", "ᵃ", "ª", "º",
    "́", "̀", "̈", "ा", "े", "न", "म", "ส", "ั", "้", "ا", "ل", "َ",
    "ש", "ָ",
    " ", " ", "᠎", " ", " ", " ", " ", " ", " ", "　", "", "​", "‍", "﻿",
    "\U0001F600", "\U0001F468‍\U0001F4BB", "\U0001F1FA\U0001F1F8", "\U0001F3F3️‍\U0001F308", "✨", "©", "™",
    "€", "£", "→", "∑", "…", "—", "“", "”", "‘", "’", "。", "，", "、",
    "\x00", "\x01", "\x7f", "\U00020000", "\U0010FFFF", "퟿", "", "�",
]


def fuzz_string(rng: random.Random, max_parts: int = 14) -> str:
    return "".join(rng.choice(FUZZ_ALPHABET) for _ in range(rng.randint(1, max_parts)))


def random_unicode_string(rng: random.Random, max_len: int = 24) -> str:
    out = []
    for _ in range(rng.randint(1, max_len)):
        r = rng.random()
        if r < 0.35:
            cp = rng.randint(0x20, 0x7E)
        elif r < 0.55:
            cp = rng.choice([0x9, 0xA, 0xD, 0x20, 0x20, 0x27, 0x2F])
        elif r < 0.8:
            cp = rng.randint(0x80, 0x2FFF)
        elif r < 0.95:
            cp = rng.randint(0x3000, 0xFFFF)
        else:
            cp = rng.randint(0x10000, 0x10FFFF)
        if 0xD800 <= cp <= 0xDFFF:
            cp = 0x4E2D
        out.append(chr(cp))
    return "".join(out)
