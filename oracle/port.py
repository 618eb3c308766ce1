"""TEST INFRASTRUCTURE ONLY: ctypes loader for the plain-C restatement (oracle/td_oracle.c).

Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import this module.
"""
from __future__ import annotations

import ctypes
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "_build" / "libtdoracle.so"


def available() -> bool:
    return LIB_PATH.exists()


class OracleError(RuntimeError):
    pass


def _lib():
    lib = ctypes.CDLL(str(LIB_PATH))
    lib.tdo_last_error.restype = ctypes.c_char_p
    lib.tdo_create.restype = ctypes.c_void_p
    lib.tdo_create.argtypes = [ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.tdo_destroy.argtypes = [ctypes.c_void_p]
    lib.tdo_split.restype = ctypes.c_int64
    lib.tdo_split.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64]
    lib.tdo_encode.restype = ctypes.c_int64
    lib.tdo_encode.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int64, ctypes.c_void_p,
                               ctypes.c_int64, ctypes.c_int]
    lib.tdo_decode.restype = ctypes.c_int64
    lib.tdo_decode.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64]
    lib.tdo_class_of_cp.argtypes = [ctypes.c_uint32]
    lib.tdo_split_variant.restype = ctypes.c_int64
    lib.tdo_split_variant.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int]
    lib.tdo_set_variant.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.tdo_set_heap_threshold.argtypes = [ctypes.c_int64]
    return lib


VARIANT_LLAMA4, VARIANT_TEKKEN, VARIANT_CL100K, VARIANT_GPT2, VARIANT_CL100K_EOS, VARIANT_QWEN2 = 0, 1, 2, 3, 4, 5


def split(data: bytes, variant: int = VARIANT_LLAMA4) -> np.ndarray:
    """Piece END offsets of the pre-tokenizer (Llama-4 pattern, or its tekken variant) over `data`."""
    lib = _lib()
    out = np.empty(max(len(data), 1), dtype=np.int64)
    n = lib.tdo_split_variant(data, len(data), out.ctypes.data, out.size, variant)
    if n < 0:
        raise OracleError(lib.tdo_last_error().decode())
    return out[:n].copy()


def class_of_cp(cp: int) -> int:
    return _lib().tdo_class_of_cp(cp)


class OracleTokenizer:
    """CPU restatement of CoreBPE for the Llama-4 split pattern (see td_oracle.c header)."""

    def __init__(self, mergeable_ranks: dict[bytes, int], variant: int = VARIANT_LLAMA4):
        self._lib = _lib()
        items = list(mergeable_ranks.items())
        ranks = np.asarray([r for _, r in items], dtype=np.int32)
        offs = np.zeros(len(items) + 1, dtype=np.int64)
        np.cumsum([len(b) for b, _ in items], out=offs[1:])
        blob = np.frombuffer(b"".join(b for b, _ in items) or b"\0", dtype=np.uint8).copy()
        self._h = self._lib.tdo_create(len(items), blob.ctypes.data, offs.ctypes.data, ranks.ctypes.data)
        self._lib.tdo_set_variant(self._h, variant)

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.tdo_destroy(self._h)
            self._h = None

    def _enc(self, data: bytes, ordinary: int) -> np.ndarray:
        out = np.empty(max(len(data), 1), dtype=np.int32)
        n = self._lib.tdo_encode(self._h, data, len(data), out.ctypes.data, out.size, ordinary)
        if n < 0:
            raise OracleError(self._lib.tdo_last_error().decode())
        return out[:n].copy()

    def encode(self, data: bytes) -> np.ndarray:
        return self._enc(data, 0)

    def merge_piece(self, piece: bytes) -> np.ndarray:
        """byte_pair_encode of one piece as given (tiktoken.cpp:371-378): no split, no whole-piece lookup."""
        out = np.empty(max(len(piece), 1), dtype=np.int32)
        self._lib.tdo_merge_piece.restype = ctypes.c_int64
        self._lib.tdo_merge_piece.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64]
        n = self._lib.tdo_merge_piece(self._h, piece, len(piece), out.ctypes.data, out.size)
        if n < 0:
            raise OracleError(self._lib.tdo_last_error().decode())
        return out[:n].copy()

    def encode_ordinary(self, data: bytes) -> np.ndarray:
        return self._enc(data, 1)

    def encode_batch(self, text: bytes, doc_offsets) -> tuple[np.ndarray, np.ndarray]:
        toks, offs = [], [0]
        for d in range(len(doc_offsets) - 1):
            t = self.encode(bytes(text[int(doc_offsets[d]):int(doc_offsets[d + 1])]))
            toks.append(t)
            offs.append(offs[-1] + len(t))
        return (np.concatenate(toks) if toks else np.empty(0, np.int32)), np.asarray(offs, dtype=np.int64)

    def decode_bytes(self, tokens) -> bytes:
        t = np.ascontiguousarray(tokens, dtype=np.int32)
        cap = max(1, 128 * len(t))
        out = np.empty(cap, dtype=np.uint8)
        n = self._lib.tdo_decode(self._h, t.ctypes.data, len(t), out.ctypes.data, cap)
        if n < 0:
            raise OracleError(self._lib.tdo_last_error().decode())
        return out[:n].tobytes()


def set_heap_threshold(n: int) -> None:
    """Pieces longer than n bytes are merged with the O(n log n) heap form of the merge loop (default 4096); 0 = always,
    a huge value = never (the reference's own quadratic loop)."""
    _lib().tdo_set_heap_threshold(int(n))
