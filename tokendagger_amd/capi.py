"""ctypes view of the C ABI in include/tokendagger_hip.h (libtokendagger_hip.so).

This is the array-in / array-out surface used by bench.py, the GPU parity tests and the bulk
methods of `tokendagger_amd.Tokenizer`.  It performs no tokenization itself: every call goes to the
HIP library, and importing/constructing fails loudly when the library or a HIP device is missing.
"""
from __future__ import annotations

import ctypes
from pathlib import Path

import numpy as np

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "libtokendagger_hip.so"

TD_OK = 0
TD_E_CAPACITY = 5
TD_MODE_ENCODE = 0
TD_MODE_ORDINARY = 1
TD_INFO_N_PAIRS, TD_INFO_MERGE_CLOSED, TD_INFO_MAX_ID, TD_INFO_TILE_BYTES = 1, 2, 3, 4
TD_INFO_WORKSPACE_BYTES, TD_INFO_N_SPECIAL, TD_INFO_LONG_PIECES, TD_INFO_FAR_PIECES = 5, 6, 7, 8
TD_OPT_LONG_POOL_BYTES = 1
TD_OPT_PROFILE = 2
TD_OPT_PIPE_CHUNK_BYTES = 3
TD_OPT_PIPE_THREADS = 4
TD_OPT_SMALL_PATH = 5
TD_OPT_FUSED = 6
TD_OPT_GRAPH = 7
TD_OPT_DEVICE_SPECIALS = 8
TD_OPT_DIRECT = 9
TD_OPT_PACK_SPLIT = 10
TD_OPT_DEDUPE = 11
TD_OPT_OVERLAP = 12
TD_OPT_GIANT_COOP_MIN = 13
TD_OPT_SPARSE = 14
TD_INFO_DEFERRED_TILES, TD_INFO_FLAGGED_TILES = 9, 10
TD_INFO_DIRECT_TILES = 11
TD_INFO_LB_TIMEOUTS = 12
TD_INFO_REPEATS, TD_INFO_LISTED_PIECES, TD_INFO_CHAR_SEEDS = 13, 14, 15
TD_INFO_SPARSE = 16

EXPORTS = [
    "td_create", "td_clone", "td_destroy", "td_last_error", "td_encode_batch", "td_encode_device", "td_reserve",
    "td_device_status", "td_decode_bytes", "td_encode_with_special", "td_info", "td_set_option",
    "td_special_count", "td_special_get", "td_profile_read",
    "td_vocab_create", "td_vocab_destroy", "td_vocab_error", "td_vocab_load_tiktoken", "td_vocab_load_hf_special",
    "td_vocab_load_tekken", "td_vocab_load_json", "td_vocab_set_pattern", "td_vocab_pattern", "td_vocab_arrays",
    "td_create_from_vocab", "td_token_bytes", "td_single_token", "td_decode_device", "td_decode_batch", "td_encode_batch_with_special",
    "td_encode_with_special_strs", "td_encode_batch_with_special_strs", "td_profile_read_ex", "td_profile_segment_name",
    "td_comm_unique_id", "td_comm_create", "td_comm_destroy", "td_comm_gather_counts", "td_comm_bases", "td_comm_gather_tokens",
    "td_comm_last_error", "td_encode_device_with_special",
]


class TokenDaggerHipError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(message)
        self.code = code


_lib = None


def _share_hip_runtime_with_torch():
    """One process must not hold two HIP runtimes: PyTorch-ROCm wheels bundle their own libamdhip64.so.7 +
    libhsa-runtime64, and a second copy (the system one this library is linked against) cannot see the GPU
    once the first has claimed it.  When torch is installed, load ITS runtime first, globally, so that our
    DT_NEEDED libamdhip64.so.7 binds to it.  Set TOKENDAGGER_NO_TORCH=1 to use the system ROCm runtime."""
    import os
    if os.environ.get("TOKENDAGGER_NO_TORCH") == "1":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        hip = Path(list(spec.submodule_search_locations)[0]) / "lib" / "libamdhip64.so"
        if hip.exists():
            import torch  # noqa: F401  (loads the bundled runtime with the right rpaths)
            ctypes.CDLL(str(hip), mode=ctypes.RTLD_GLOBAL)
    except Exception:  # torch broken or absent: fall back to the system runtime
        return


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    import os
    lib_path = Path(os.environ.get("TD_HIP_LIB", str(LIB_PATH)))  # TD_HIP_LIB: kernel-tuning builds only
    if not lib_path.exists():
        raise ImportError(
            f"{lib_path} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'); "
            "tokendagger_amd has no CPU fallback")
    _share_hip_runtime_with_torch()
    lib = ctypes.CDLL(str(lib_path), mode=ctypes.RTLD_GLOBAL)
    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
    lib.td_encode_device_with_special.restype = i32
    lib.td_encode_device_with_special.argtypes = [vp, vp, i64, vp, i64, vp, i64, vp, i64, vp, vp]
    lib.td_comm_unique_id.restype = i32
    lib.td_comm_unique_id.argtypes = [vp]
    lib.td_comm_create.restype = i32
    lib.td_comm_create.argtypes = [vp, i32, i32, i32, ctypes.POINTER(vp)]
    lib.td_comm_destroy.argtypes = [vp]
    lib.td_comm_gather_counts.restype = i32
    lib.td_comm_gather_counts.argtypes = [vp, vp, vp, vp]
    lib.td_comm_bases.restype = i32
    lib.td_comm_bases.argtypes = [vp, i32, i32, ctypes.POINTER(i64), ctypes.POINTER(i64), ctypes.POINTER(i64), ctypes.POINTER(i64)]
    lib.td_comm_gather_tokens.restype = i32
    lib.td_comm_gather_tokens.argtypes = [vp, vp, vp, i32, vp, i64, vp]
    lib.td_comm_last_error.restype = ctypes.c_char_p
    lib.td_create.restype = i32
    lib.td_create.argtypes = [ctypes.c_char_p, i64, vp, vp, vp, i64, vp, vp, vp, i32, ctypes.POINTER(vp)]
    lib.td_destroy.argtypes = [vp]
    lib.td_clone.restype = i32
    lib.td_clone.argtypes = [vp, ctypes.POINTER(vp)]
    lib.td_last_error.restype = ctypes.c_char_p
    lib.td_last_error.argtypes = [vp]
    lib.td_encode_batch.restype = i32
    lib.td_encode_batch.argtypes = [vp, vp, vp, i64, i32, vp, i64, vp, ctypes.POINTER(i64)]
    lib.td_encode_device.restype = i32
    lib.td_encode_device.argtypes = [vp, vp, i64, vp, i64, i32, vp, i64, vp, vp]
    lib.td_reserve.restype = i32
    lib.td_reserve.argtypes = [vp, i64, i64]
    lib.td_device_status.restype = i32
    lib.td_device_status.argtypes = [vp, vp, ctypes.POINTER(i64)]
    lib.td_decode_bytes.restype = i32
    lib.td_decode_bytes.argtypes = [vp, vp, i64, vp, i64, ctypes.POINTER(i64)]
    lib.td_encode_with_special.restype = i32
    lib.td_encode_with_special.argtypes = [vp, vp, i64, vp, i64, vp, i64, ctypes.POINTER(i64), ctypes.POINTER(ctypes.c_int32)]
    lib.td_info.restype = i64
    lib.td_info.argtypes = [vp, i32]
    lib.td_set_option.restype = i32
    lib.td_set_option.argtypes = [vp, i32, i64]
    lib.td_profile_read.restype = i32
    lib.td_profile_read.argtypes = [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(i64)]
    lib.td_profile_read_ex.restype = i32
    lib.td_profile_read_ex.argtypes = [vp, ctypes.POINTER(ctypes.c_double), i32, ctypes.POINTER(i64)]
    lib.td_profile_segment_name.restype = ctypes.c_char_p
    lib.td_profile_segment_name.argtypes = [i32]
    lib.td_special_count.restype = i64
    lib.td_special_count.argtypes = [vp]
    lib.td_special_get.restype = i32
    lib.td_special_get.argtypes = [vp, i64, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(i64), ctypes.POINTER(ctypes.c_int32)]
    lib.td_encode_batch_with_special.restype = i32
    lib.td_encode_batch_with_special.argtypes = [vp, vp, vp, i64, vp, i64, vp, i64, vp, ctypes.POINTER(i64)]
    lib.td_encode_with_special_strs.restype = i32
    lib.td_encode_with_special_strs.argtypes = [vp, vp, i64, vp, vp, i64, vp, i64, ctypes.POINTER(i64), ctypes.POINTER(ctypes.c_int32)]
    lib.td_encode_batch_with_special_strs.restype = i32
    lib.td_encode_batch_with_special_strs.argtypes = [vp, vp, vp, i64, vp, vp, i64, vp, i64, vp, ctypes.POINTER(i64)]
    lib.td_decode_batch.restype = i32
    lib.td_decode_batch.argtypes = [vp, vp, vp, i64, vp, i64, vp, ctypes.POINTER(i64)]
    lib.td_decode_device.restype = i32
    lib.td_decode_device.argtypes = [vp, vp, i64, vp, i64, vp, vp]
    lib.td_token_bytes.restype = i32
    lib.td_token_bytes.argtypes = [vp, ctypes.c_int32, ctypes.POINTER(vp), ctypes.POINTER(i64)]
    lib.td_single_token.restype = i32
    lib.td_single_token.argtypes = [vp, vp, i64, ctypes.POINTER(ctypes.c_int32)]
    lib.td_vocab_create.restype = i32
    lib.td_vocab_create.argtypes = [ctypes.POINTER(vp)]
    lib.td_vocab_destroy.argtypes = [vp]
    lib.td_vocab_error.restype = ctypes.c_char_p
    lib.td_vocab_error.argtypes = [vp]
    for fn in ("td_vocab_load_tiktoken", "td_vocab_load_tekken", "td_vocab_set_pattern"):
        getattr(lib, fn).restype = i32
        getattr(lib, fn).argtypes = [vp, ctypes.c_char_p]
    lib.td_vocab_load_hf_special.restype = i32
    lib.td_vocab_load_hf_special.argtypes = [vp, ctypes.c_char_p, i32]
    lib.td_vocab_load_json.restype = i32
    lib.td_vocab_load_json.argtypes = [vp, ctypes.c_char_p, ctypes.c_char_p]
    lib.td_vocab_pattern.restype = ctypes.c_char_p
    lib.td_vocab_pattern.argtypes = [vp]
    lib.td_vocab_arrays.restype = i32
    lib.td_vocab_arrays.argtypes = [vp, i32, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(i64)]
    lib.td_create_from_vocab.restype = i32
    lib.td_create_from_vocab.argtypes = [vp, i32, ctypes.POINTER(vp)]
    _lib = lib
    return lib


def _pack(items: list[tuple[bytes, int]]):
    ranks = np.asarray([r for _, r in items], dtype=np.int32)
    offs = np.zeros(len(items) + 1, dtype=np.int64)
    if items:
        np.cumsum([len(b) for b, _ in items], out=offs[1:])
    blob = np.frombuffer(b"".join(b for b, _ in items) or b"\0", dtype=np.uint8).copy()
    return blob, offs, ranks


def _as_u8(data) -> np.ndarray:
    if isinstance(data, np.ndarray):
        return np.ascontiguousarray(data, dtype=np.uint8)
    return np.frombuffer(data, dtype=np.uint8) if len(data) else np.zeros(0, dtype=np.uint8)


class Vocab:
    """A `td_vocab`: vocabulary files read by the C++ loaders (tiktoken .model, HF tokenizer_config.json,
    tekken.json, the reference wrapper's JSON files).  Host only: works without a GPU."""

    def __init__(self):
        self._lib = load_library()
        h = ctypes.c_void_p()
        if self._lib.td_vocab_create(ctypes.byref(h)) != TD_OK:
            raise TokenDaggerHipError(1, "td_vocab_create failed")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.td_vocab_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc: int):
        if rc != TD_OK:
            raise TokenDaggerHipError(rc, self._lib.td_vocab_error(self._h).decode("utf-8", "replace"))
        return self

    @staticmethod
    def _p(path) -> bytes:
        import os
        return os.fsencode(str(path))

    def load_tiktoken(self, path):
        return self._check(self._lib.td_vocab_load_tiktoken(self._h, self._p(path)))

    def load_hf_special(self, path, also_mergeable: bool = False):
        return self._check(self._lib.td_vocab_load_hf_special(self._h, self._p(path), int(also_mergeable)))

    def load_tekken(self, path):
        return self._check(self._lib.td_vocab_load_tekken(self._h, self._p(path)))

    def load_json(self, vocab_path=None, special_path=None):
        return self._check(self._lib.td_vocab_load_json(self._h, self._p(vocab_path) if vocab_path else None,
                                                        self._p(special_path) if special_path else None))

    def set_pattern(self, pat_str: str):
        return self._check(self._lib.td_vocab_set_pattern(self._h, pat_str.encode("utf-8")))

    @property
    def pattern(self) -> str:
        return self._lib.td_vocab_pattern(self._h).decode("utf-8")

    def arrays(self, special: bool = False):
        """-> (bytes uint8[total], offsets int64[n+1], ranks int32[n]) copies of the loaded tokens."""
        b, o, r = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        n = ctypes.c_int64(0)
        self._check(self._lib.td_vocab_arrays(self._h, int(special), ctypes.byref(b), ctypes.byref(o), ctypes.byref(r),
                                              ctypes.byref(n)))
        cnt = n.value
        offs = np.ctypeslib.as_array(ctypes.cast(o, ctypes.POINTER(ctypes.c_int64)), shape=(cnt + 1,)).copy()
        ranks = (np.ctypeslib.as_array(ctypes.cast(r, ctypes.POINTER(ctypes.c_int32)), shape=(cnt,)).copy()
                 if cnt else np.zeros(0, dtype=np.int32))
        total = int(offs[-1])
        blob = (np.ctypeslib.as_array(ctypes.cast(b, ctypes.POINTER(ctypes.c_uint8)), shape=(total,)).copy()
                if total else np.zeros(0, dtype=np.uint8))
        return blob, offs, ranks

    def __len__(self) -> int:
        n = ctypes.c_int64(0)
        self._lib.td_vocab_arrays(self._h, 0, None, None, None, ctypes.byref(n))
        return n.value

    def mergeable_ranks(self) -> dict[bytes, int]:
        blob, offs, ranks = self.arrays(False)
        raw = blob.tobytes()
        return {raw[offs[i]:offs[i + 1]]: int(ranks[i]) for i in range(len(ranks))}

    def special_tokens(self) -> dict[str, int]:
        blob, offs, ranks = self.arrays(True)
        raw = blob.tobytes()
        return {raw[offs[i]:offs[i + 1]].decode("utf-8"): int(ranks[i]) for i in range(len(ranks))}


def load_tiktoken_bpe(path) -> dict[bytes, int]:
    """tiktoken.load.load_tiktoken_bpe for a local file, through the C++ loader."""
    return Vocab().load_tiktoken(path).mergeable_ranks()


class HipTokenizer:
    """Owns one `td_tokenizer` handle (device tables + workspace on one GPU)."""

    @classmethod
    def from_vocab(cls, vocab: Vocab, device: int = -1) -> "HipTokenizer":
        """td_create_from_vocab: no Python-side token objects at all."""
        self = cls.__new__(cls)
        self._lib = load_library()
        self._h = None
        h = ctypes.c_void_p()
        rc = self._lib.td_create_from_vocab(vocab._h, device, ctypes.byref(h))
        if rc != TD_OK:
            raise TokenDaggerHipError(rc, self._lib.td_last_error(None).decode("utf-8", "replace"))
        self._h = h
        return self

    def __init__(self, pat_str: str, mergeable_ranks: dict[bytes, int], special_tokens: dict[str, int] | None = None,
                 device: int = -1):
        self._lib = load_library()
        self._h = None
        special_tokens = special_tokens or {}
        b, o, r = _pack(list(mergeable_ranks.items()))
        sb, so, sr = _pack([(k.encode("utf-8"), v) for k, v in special_tokens.items()])
        h = ctypes.c_void_p()
        rc = self._lib.td_create(pat_str.encode("utf-8"), len(r), b.ctypes.data, o.ctypes.data, r.ctypes.data,
                                 len(sr), sb.ctypes.data, so.ctypes.data, sr.ctypes.data, device, ctypes.byref(h))
        if rc != TD_OK:
            raise TokenDaggerHipError(rc, self._lib.td_last_error(None).decode("utf-8", "replace"))
        self._h = h

    def clone(self) -> "HipTokenizer":
        """td_clone: a second handle on the same device tables (own lock, workspace and streams) — one per host thread / HIP
        stream for concurrent encodes.  Either handle may be closed first."""
        other = type(self).__new__(type(self))
        other._lib = self._lib
        other._h = None
        h = ctypes.c_void_p()
        rc = self._lib.td_clone(self._h, ctypes.byref(h))
        if rc != TD_OK:
            raise TokenDaggerHipError(rc, self._lib.td_last_error(None).decode("utf-8", "replace"))
        other._h = h
        return other

    def close(self):
        if getattr(self, "_h", None):
            self._lib.td_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc: int):
        if rc != TD_OK:
            raise TokenDaggerHipError(rc, self._lib.td_last_error(self._h).decode("utf-8", "replace"))

    # ---- host-buffer API --------------------------------------------------------------------
    def encode_batch(self, text, doc_offsets, mode: int = TD_MODE_ENCODE, capacity: int | None = None):
        """text: bytes / uint8 array of all documents concatenated; doc_offsets: int64[n_docs+1].
        -> (tokens int32[total], offsets int64[n_docs+1])"""
        buf = _as_u8(text)
        offs = np.ascontiguousarray(doc_offsets, dtype=np.int64)
        n_docs = len(offs) - 1
        n = int(offs[-1]) if len(offs) else 0
        cap = capacity if capacity is not None else max(16, n // 3 + 16)
        out_offs = np.empty(n_docs + 1, dtype=np.int64)
        ntok = ctypes.c_int64(0)
        for _ in range(2):
            toks = np.empty(cap, dtype=np.int32)
            rc = self._lib.td_encode_batch(self._h, buf.ctypes.data if n else None, offs.ctypes.data, n_docs, mode,
                                           toks.ctypes.data, cap, out_offs.ctypes.data, ctypes.byref(ntok))
            if rc == TD_E_CAPACITY and capacity is None and ntok.value > cap:
                cap = ntok.value
                continue
            break
        self._check(rc)
        return toks[:ntok.value], out_offs

    def encode(self, data, mode: int = TD_MODE_ENCODE) -> np.ndarray:
        buf = _as_u8(data)
        toks, _ = self.encode_batch(buf, np.asarray([0, len(buf)], dtype=np.int64), mode)
        return toks

    def encode_with_special(self, data, allowed_ids) -> tuple[np.ndarray, int]:
        buf = _as_u8(data)
        ids = np.ascontiguousarray(sorted(allowed_ids), dtype=np.int32)
        cap = len(buf) + 16
        toks = np.empty(cap, dtype=np.int32)
        ntok = ctypes.c_int64(0)
        last = ctypes.c_int32(0)
        rc = self._lib.td_encode_with_special(self._h, buf.ctypes.data if len(buf) else None, len(buf),
                                              ids.ctypes.data if len(ids) else None, len(ids), toks.ctypes.data, cap,
                                              ctypes.byref(ntok), ctypes.byref(last))
        self._check(rc)
        return toks[:ntok.value].copy(), last.value

    @staticmethod
    def _pack_strs(strs):
        enc = [s.encode("utf-8") if isinstance(s, str) else bytes(s) for s in strs]
        offs = np.zeros(len(enc) + 1, dtype=np.int64)
        if enc:
            np.cumsum([len(b) for b in enc], out=offs[1:])
        blob = np.frombuffer(b"".join(enc) or b"\0", dtype=np.uint8).copy()
        return blob, offs

    def encode_with_special_strs(self, data, allowed) -> tuple[np.ndarray, int]:
        """allowed: the special-token STRINGS that may be cut out (tiktoken's allowed_special)."""
        buf = _as_u8(data)
        ab, ao = self._pack_strs(list(allowed))
        cap = len(buf) + 16
        toks = np.empty(cap, dtype=np.int32)
        ntok = ctypes.c_int64(0)
        last = ctypes.c_int32(0)
        rc = self._lib.td_encode_with_special_strs(self._h, buf.ctypes.data if len(buf) else None, len(buf), ab.ctypes.data,
                                                   ao.ctypes.data, len(ao) - 1, toks.ctypes.data, cap, ctypes.byref(ntok),
                                                   ctypes.byref(last))
        self._check(rc)
        return toks[:ntok.value].copy(), last.value

    def encode_batch_with_special_strs(self, text, doc_offsets, allowed):
        buf = _as_u8(text)
        offs = np.ascontiguousarray(doc_offsets, dtype=np.int64)
        ab, ao = self._pack_strs(list(allowed))
        n_docs = len(offs) - 1
        n = int(offs[-1]) if len(offs) else 0
        cap = n + 16
        out_offs = np.empty(n_docs + 1, dtype=np.int64)
        toks = np.empty(cap, dtype=np.int32)
        ntok = ctypes.c_int64(0)
        rc = self._lib.td_encode_batch_with_special_strs(self._h, buf.ctypes.data if n else None, offs.ctypes.data, n_docs,
                                                         ab.ctypes.data, ao.ctypes.data, len(ao) - 1, toks.ctypes.data, cap,
                                                         out_offs.ctypes.data, ctypes.byref(ntok))
        self._check(rc)
        return toks[:ntok.value].copy(), out_offs

    def decode_bytes(self, tokens) -> bytes:
        t = np.ascontiguousarray(tokens, dtype=np.int32)
        nb = ctypes.c_int64(0)
        cap = max(64, 8 * len(t))
        for _ in range(2):
            out = np.empty(cap, dtype=np.uint8)
            rc = self._lib.td_decode_bytes(self._h, t.ctypes.data if len(t) else None, len(t), out.ctypes.data, cap,
                                           ctypes.byref(nb))
            if rc == TD_E_CAPACITY and nb.value > cap:
                cap = nb.value
                continue
            break
        self._check(rc)
        return out[:nb.value].tobytes()

    # ---- device-buffer API (pointers are raw device addresses, e.g. torch_tensor.data_ptr()) --
    def encode_batch_with_special(self, text, doc_offsets, allowed_ids):
        """encode_batch with allowed special tokens (ids): -> (tokens int32[total], offsets int64[n_docs+1])"""
        buf = _as_u8(text)
        offs = np.ascontiguousarray(doc_offsets, dtype=np.int64)
        ids = np.ascontiguousarray(sorted(allowed_ids), dtype=np.int32)
        n_docs = len(offs) - 1
        n = int(offs[-1]) if len(offs) else 0
        cap = max(16, n // 3 + 16)
        out_offs = np.empty(n_docs + 1, dtype=np.int64)
        ntok = ctypes.c_int64(0)
        for _ in range(2):
            toks = np.empty(cap, dtype=np.int32)
            rc = self._lib.td_encode_batch_with_special(self._h, buf.ctypes.data if n else None, offs.ctypes.data, n_docs,
                                                        ids.ctypes.data if len(ids) else None, len(ids), toks.ctypes.data, cap,
                                                        out_offs.ctypes.data, ctypes.byref(ntok))
            if rc == TD_E_CAPACITY and ntok.value > cap:
                cap = ntok.value
                continue
            break
        self._check(rc)
        return toks[:ntok.value], out_offs

    def decode_batch(self, tokens, tok_offsets) -> tuple[bytes, np.ndarray]:
        """ids of all documents concatenated + int64 offsets -> (all bytes concatenated, int64 byte offsets)."""
        t = np.ascontiguousarray(tokens, dtype=np.int32)
        o = np.ascontiguousarray(tok_offsets, dtype=np.int64)
        n_docs = len(o) - 1
        out_offs = np.empty(n_docs + 1, dtype=np.int64)
        cap = max(64, 8 * len(t))
        nb = ctypes.c_int64(0)
        for _ in range(2):
            out = np.empty(cap, dtype=np.uint8)
            rc = self._lib.td_decode_batch(self._h, t.ctypes.data if len(t) else None, o.ctypes.data, n_docs, out.ctypes.data, cap,
                                           out_offs.ctypes.data, ctypes.byref(nb))
            if rc == TD_E_CAPACITY and nb.value > cap:
                cap = nb.value
                continue
            break
        self._check(rc)
        return out[:nb.value].tobytes(), out_offs

    def decode_device(self, d_tokens: int, n_tokens: int, d_out: int, out_capacity: int, d_n_bytes: int = 0, stream: int = 0):
        """Device pointers in, asynchronous on `stream`; check with device_status(stream)."""
        self._check(self._lib.td_decode_device(self._h, d_tokens, n_tokens, d_out, out_capacity, d_n_bytes or None, stream or None))

    def reserve(self, max_bytes: int, max_docs: int):
        self._check(self._lib.td_reserve(self._h, max_bytes, max_docs))

    def encode_device(self, d_text: int, n_bytes: int, d_doc_offsets: int, n_docs: int, d_out_tokens: int,
                      out_capacity: int, d_out_offsets: int, stream: int = 0, mode: int = TD_MODE_ENCODE):
        """Asynchronous on `stream`; d_out_offsets[n_docs] receives the total token count."""
        self._check(self._lib.td_encode_device(self._h, d_text, n_bytes, d_doc_offsets, n_docs, mode, d_out_tokens,
                                               out_capacity, d_out_offsets, stream))

    def encode_device_with_special(self, d_text: int, n_bytes: int, d_doc_offsets: int, n_docs: int, allowed_ids, d_out_tokens: int,
                                   out_capacity: int, d_out_offsets: int, stream: int = 0):
        """td_encode_device_with_special: allowed special tokens (ids) are searched for and cut out ON THE DEVICE."""
        ids = np.ascontiguousarray(np.asarray(sorted(set(int(i) for i in allowed_ids)), dtype=np.int32))
        self._check(self._lib.td_encode_device_with_special(self._h, ctypes.c_void_p(d_text), n_bytes, ctypes.c_void_p(d_doc_offsets), n_docs,
                                                            ids.ctypes.data_as(ctypes.c_void_p), len(ids), ctypes.c_void_p(d_out_tokens),
                                                            out_capacity, ctypes.c_void_p(d_out_offsets), ctypes.c_void_p(stream)))

    def device_status(self, stream: int = 0):
        pos = ctypes.c_int64(0)
        self._check(self._lib.td_device_status(self._h, stream, ctypes.byref(pos)))

    # ---- misc -------------------------------------------------------------------------------
    def info(self, what: int) -> int:
        return int(self._lib.td_info(self._h, what))

    def set_option(self, what: int, value: int):
        self._check(self._lib.td_set_option(self._h, what, value))

    def profile_read(self) -> tuple[float, float, int]:
        """-> (sum of td_split_tiles ms, sum of td_encode_tiles ms, calls) since the last read (TD_OPT_PROFILE=1)."""
        a = ctypes.c_double(0); b = ctypes.c_double(0); n = ctypes.c_int64(0)
        self._check(self._lib.td_profile_read(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(n)))
        return a.value, b.value, n.value

    def profile_read_all(self) -> tuple[dict[str, float], int]:
        """-> ({kernel segment: summed ms}, calls) since the last read (TD_OPT_PROFILE=1)."""
        names = []
        while True:
            nm = self._lib.td_profile_segment_name(len(names)).decode()
            if not nm:
                break
            names.append(nm)
        arr = (ctypes.c_double * len(names))()
        n = ctypes.c_int64(0)
        self._check(self._lib.td_profile_read_ex(self._h, arr, len(names), ctypes.byref(n)))
        return {nm: arr[i] for i, nm in enumerate(names)}, n.value

    def special_tokens(self) -> dict[str, int]:
        out = {}
        for i in range(self._lib.td_special_count(self._h)):
            s = ctypes.c_char_p(); n = ctypes.c_int64(0); tid = ctypes.c_int32(0)
            self._lib.td_special_get(self._h, i, ctypes.byref(s), ctypes.byref(n), ctypes.byref(tid))
            out[ctypes.string_at(s, n.value).decode("utf-8")] = tid.value
        return out


# ---------------------------------------------------------------------------------------------- multi-GPU epilogue
TD_COMM_ID_BYTES = 128


def comm_bases(table, rank: int) -> tuple[int, int, int, int]:
    """td_comm_bases: gathered {tokens, documents} per rank (host array of 2 * world int64) -> (token_base, doc_base,
    token_total, doc_total) of `rank`.  Host only: no device, no RCCL."""
    lib = load_library()
    t = np.ascontiguousarray(np.asarray(table, dtype=np.int64).reshape(-1))
    world = len(t) // 2
    tb, db, tt, dt = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    rc = lib.td_comm_bases(t.ctypes.data_as(ctypes.c_void_p), world, rank, ctypes.byref(tb), ctypes.byref(db), ctypes.byref(tt), ctypes.byref(dt))
    if rc != 0:
        raise TokenDaggerHipError(rc, (lib.td_comm_last_error() or b"").decode("utf-8", "replace"))
    return tb.value, db.value, tt.value, dt.value


def comm_unique_id() -> bytes:
    """td_comm_unique_id (rank 0): the 128 bytes every rank hands to RcclComm."""
    lib = load_library()
    buf = (ctypes.c_uint8 * TD_COMM_ID_BYTES)()
    rc = lib.td_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p))
    if rc != 0:
        raise TokenDaggerHipError(rc, (lib.td_comm_last_error() or b"").decode("utf-8", "replace"))
    return bytes(buf)


class RcclComm:
    """The path's only exchange behind the C ABI (td_comm_*): RCCL all-gather of {tokens, documents}, optional gather of the
    ids to one rank.  Pointers are raw device addresses (e.g. torch.Tensor.data_ptr())."""

    def __init__(self, unique_id: bytes, world: int, rank: int, device: int = -1):
        self._lib = load_library()
        self._h = ctypes.c_void_p()
        self.world, self.rank = world, rank
        idb = (ctypes.c_uint8 * TD_COMM_ID_BYTES).from_buffer_copy(unique_id)
        self._check(self._lib.td_comm_create(ctypes.cast(idb, ctypes.c_void_p), world, rank, device, ctypes.byref(self._h)))

    def _check(self, rc: int):
        if rc != 0:
            raise TokenDaggerHipError(rc, (self._lib.td_comm_last_error() or b"").decode("utf-8", "replace"))

    def gather_counts(self, d_counts: int, d_table: int, stream: int = 0):
        self._check(self._lib.td_comm_gather_counts(self._h, ctypes.c_void_p(d_counts), ctypes.c_void_p(d_table), ctypes.c_void_p(stream)))

    def gather_tokens(self, d_tokens: int, table, root: int, d_root_tokens: int, root_capacity: int, stream: int = 0):
        t = np.ascontiguousarray(np.asarray(table, dtype=np.int64).reshape(-1))
        self._check(self._lib.td_comm_gather_tokens(self._h, ctypes.c_void_p(d_tokens), t.ctypes.data_as(ctypes.c_void_p), root,
                                                    ctypes.c_void_p(d_root_tokens), root_capacity, ctypes.c_void_p(stream)))

    def close(self):
        if self._h:
            self._lib.td_comm_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass
