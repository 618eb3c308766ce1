"""TEST INFRASTRUCTURE ONLY: ctypes loader for oracle/_ref/libtdref.so (the compiled reference).

Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import this module.
"""
from __future__ import annotations

import ctypes
import os
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "_ref" / "libtdref.so"

_u8p = ctypes.POINTER(ctypes.c_uint8)
_i32p = ctypes.POINTER(ctypes.c_int32)
_i64p = ctypes.POINTER(ctypes.c_int64)


def available() -> bool:
    return LIB_PATH.exists()


INTERP_PATH = LIB_PATH.parent / "libpcre2interp.so"


def interp_available() -> bool:
    return INTERP_PATH.exists()


_interp = None


def interp_split(pattern: str, data: bytes) -> list[bytes]:
    """The reference's split loop on PCRE2's INTERPRETER (oracle/pcre2_interp.c; the reference itself uses the JIT)."""
    global _interp
    if _interp is None:
        _interp = ctypes.CDLL(str(INTERP_PATH))
        _interp.interp_split.restype = ctypes.c_int64
        _interp.interp_split.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64]
    out = np.zeros(2 * (len(data) + 2), dtype=np.int64)
    k = _interp.interp_split(pattern.encode("utf-8"), data, len(data), out.ctypes.data, len(data) + 2)
    if k < 0:
        raise ValueError("PCRE2 does not compile the pattern")
    return [data[out[2 * i]:out[2 * i + 1]] for i in range(k)]


def _lib():
    lib = ctypes.CDLL(str(LIB_PATH))
    lib.tdref_last_error.restype = ctypes.c_char_p
    lib.tdref_create.restype = ctypes.c_void_p
    lib.tdref_create.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                 ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                 ctypes.c_void_p]
    lib.tdref_destroy.argtypes = [ctypes.c_void_p]
    lib.tdref_encode.restype = ctypes.c_int64
    lib.tdref_encode.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int64, ctypes.c_void_p,
                                 ctypes.c_int64, ctypes.c_void_p]
    lib.tdref_encode_ordinary.restype = ctypes.c_int64
    lib.tdref_encode_ordinary.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int64,
                                          ctypes.c_void_p, ctypes.c_int64]
    lib.tdref_encode_special.restype = ctypes.c_int64
    lib.tdref_encode_special.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int64,
                                         ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64]
    lib.tdref_split.restype = ctypes.c_int64
    lib.tdref_split.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int64, ctypes.c_void_p,
                                ctypes.c_int64]
    lib.tdref_split_bytes.restype = ctypes.c_int64
    lib.tdref_split_bytes.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int64, ctypes.c_char_p, ctypes.c_int64,
                                      ctypes.c_void_p, ctypes.c_int64]
    lib.tdref_decode.restype = ctypes.c_int64
    lib.tdref_decode.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                 ctypes.c_int64]
    lib.tdref_time_encode_batch.restype = ctypes.c_double
    lib.tdref_time_encode_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_int64, ctypes.c_int, ctypes.c_void_p,
                                            ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
    lib.tdref_pcre2_version.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int]
    return lib


def pack_vocab(mergeable_ranks: dict[bytes, int]):
    items = list(mergeable_ranks.items())
    ranks = np.asarray([r for _, r in items], dtype=np.int32)
    offs = np.zeros(len(items) + 1, dtype=np.int64)
    np.cumsum([len(b) for b, _ in items], out=offs[1:])
    blob = np.frombuffer(b"".join(b for b, _ in items) or b"\0", dtype=np.uint8).copy()
    return blob, offs, ranks


class RefError(RuntimeError):
    pass


class RefTokenizer:
    """The reference's `tiktoken::CoreBPE` (tiktoken.hpp:38-88) behind a byte-buffer interface."""

    def __init__(self, pat_str: str, mergeable_ranks: dict[bytes, int],
                 special_tokens: dict[str, int] | None = None):
        self._lib = _lib()
        special_tokens = special_tokens or {}
        b, o, r = pack_vocab(mergeable_ranks)
        sb, so, sr = pack_vocab({k.encode("utf-8"): v for k, v in special_tokens.items()})
        self._h = self._lib.tdref_create(pat_str.encode("utf-8"), len(r), b.ctypes.data, o.ctypes.data,
                                         r.ctypes.data, len(sr), sb.ctypes.data, so.ctypes.data,
                                         sr.ctypes.data)
        if not self._h:
            raise RefError(self._lib.tdref_last_error().decode())

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.tdref_destroy(self._h)
            self._h = None

    def _err(self):
        return RefError(self._lib.tdref_last_error().decode())

    def pcre2_version(self) -> tuple[str, str]:
        a = ctypes.create_string_buffer(64); b = ctypes.create_string_buffer(64)
        self._lib.tdref_pcre2_version(a, 64, b, 64)
        return a.value.decode(), b.value.decode()

    def encode(self, data: bytes, return_last_piece_len: bool = False):
        out = np.empty(max(len(data), 1), dtype=np.int32)
        last = ctypes.c_int32(0)
        n = self._lib.tdref_encode(self._h, data, len(data), out.ctypes.data, out.size, ctypes.byref(last))
        if n < 0:
            raise self._err()
        return (out[:n].copy(), last.value) if return_last_piece_len else out[:n].copy()

    def encode_ordinary(self, data: bytes) -> np.ndarray:
        out = np.empty(max(len(data), 1), dtype=np.int32)
        n = self._lib.tdref_encode_ordinary(self._h, data, len(data), out.ctypes.data, out.size)
        if n < 0:
            raise self._err()
        return out[:n].copy()

    def encode_special(self, data: bytes, allowed: list[str]) -> np.ndarray:
        out = np.empty(max(len(data), 1), dtype=np.int32)
        arr = (ctypes.c_char_p * len(allowed))(*[a.encode("utf-8") for a in allowed])
        n = self._lib.tdref_encode_special(self._h, data, len(data), arr, len(allowed), out.ctypes.data, out.size)
        if n < 0:
            raise self._err()
        return out[:n].copy()

    def split(self, data: bytes) -> np.ndarray:
        """Piece END offsets (int64) of split_text over the whole buffer."""
        out = np.empty(max(len(data), 1), dtype=np.int64)
        n = self._lib.tdref_split(self._h, data, len(data), out.ctypes.data, out.size)
        if n < 0:
            raise self._err()
        return out[:n].copy()

    def split_pieces(self, data: bytes) -> list[bytes]:
        """The pieces of split_text themselves (a pattern may skip text: their lengths need not add up to offsets)."""
        lens = np.empty(max(len(data), 1), dtype=np.int64)
        buf = ctypes.create_string_buffer(max(len(data), 1))
        n = self._lib.tdref_split_bytes(self._h, data, len(data), buf, len(data), lens.ctypes.data, lens.size)
        if n < 0:
            raise self._err()
        raw, out, pos = buf.raw, [], 0
        for k in range(n):
            out.append(raw[pos:pos + int(lens[k])])
            pos += int(lens[k])
        return out

    def decode_bytes(self, tokens) -> bytes:
        t = np.ascontiguousarray(tokens, dtype=np.int32)
        cap = max(1, 128 * len(t))
        out = np.empty(cap, dtype=np.uint8)
        n = self._lib.tdref_decode(self._h, t.ctypes.data, len(t), out.ctypes.data, cap)
        if n < 0:
            raise self._err()
        return out[:n].tobytes()

    def encode_batch(self, text: bytes | np.ndarray, doc_offsets, n_threads: int = 1,
                     want_tokens: bool = True):
        """-> (seconds_in_encode, tokens|None, offsets) for documents text[off[d]:off[d+1]]."""
        buf = np.frombuffer(text, dtype=np.uint8) if isinstance(text, (bytes, bytearray)) else text
        offs = np.ascontiguousarray(doc_offsets, dtype=np.int64)
        n_docs = len(offs) - 1
        out_off = np.empty(n_docs + 1, dtype=np.int64)
        cap = int(offs[-1] - offs[0]) + 1
        toks = np.empty(cap, dtype=np.int32) if want_tokens else None
        ntok = ctypes.c_int64(0)
        sec = self._lib.tdref_time_encode_batch(self._h, buf.ctypes.data, offs.ctypes.data, n_docs,
                                                int(n_threads), toks.ctypes.data if want_tokens else None,
                                                cap, out_off.ctypes.data, ctypes.byref(ntok))
        if sec < 0:
            raise self._err()
        return sec, (toks[:ntok.value] if want_tokens else None), out_off
