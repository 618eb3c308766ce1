"""tiktoken-compatible Python surface over the MI355X tokenizer core.

Mirrors the public interface of the reference's ``tokendagger`` package (class / function names,
keyword arguments, defaults, exception types: /root/reference/tokendagger/wrapper.py:28-395) so that
``import tokendagger as tiktoken`` keeps working, but every encode/decode goes to the HIP library
through ``_tokendagger_core`` — there is no CPU tokenization in this package.  Batch methods hand the
whole batch to the GPU in one call instead of fanning single calls out to a thread pool
(reference: wrapper.py:212-235); ``num_threads`` is accepted for compatibility and ignored.
"""
from __future__ import annotations

import json
from pathlib import Path
from typing import AbstractSet, Collection, Literal, Sequence

import numpy as np

from . import capi as _capi

_capi.load_library()  # (first: it makes the HIP runtime of this process the one torch bundles, when torch is installed)
from . import _tokendagger_core as _core  # noqa: E402

MODE_ENCODE, MODE_ORDINARY = 0, 1


class TokenDaggerError(Exception):
    """Base exception for TokenDagger errors (reference: wrapper.py:23-25)."""


def _vocab_items(vocab) -> list:
    """Accepts the reference's list-of-dicts form ({'rank','token_bytes','token_string'}) or a tiktoken
    ``mergeable_ranks`` dict {bytes: rank}."""
    items = []
    if isinstance(vocab, dict):
        for token_bytes, rank in vocab.items():
            it = _core.VocabItem()
            it.rank = int(rank)
            it.token_bytes = list(token_bytes)
            items.append(it)
        return items
    for entry in vocab:
        it = _core.VocabItem()
        it.rank = int(entry["rank"])
        it.token_bytes = list(entry["token_bytes"])
        it.token_string = entry.get("token_string", "")
        items.append(it)
    return items


class Tokenizer:
    """High-level tokenizer with the tiktoken ``Encoding`` methods (reference: wrapper.py:28-326)."""

    def __init__(
        self,
        name: str,
        *,
        pattern: str | None = None,
        pat_str: str | None = None,
        vocab: list[dict] | None = None,
        mergeable_ranks: dict[bytes, int] | None = None,
        special_tokens: dict[str, int] | None = None,
        vocab_file: str | Path | None = None,
        special_tokens_file: str | Path | None = None,
        device: int = -1,
    ):
        self.name = name
        self.pattern = pat_str if pat_str is not None else pattern
        if mergeable_ranks is not None:
            vocab = mergeable_ranks
        if vocab_file:
            vocab = self._read_json(vocab_file, "Vocabulary")
        elif vocab is None:
            raise ValueError("Either 'vocab', 'mergeable_ranks', or 'vocab_file' must be provided")
        if special_tokens_file:
            special_tokens = self._read_json(special_tokens_file, "Special tokens")
        elif special_tokens is None:
            special_tokens = {}
        self._special_tokens = dict(special_tokens)
        ranks = list(vocab.values()) if isinstance(vocab, dict) else [e["rank"] for e in vocab]
        self.max_token_value = max(max(ranks), max(self._special_tokens.values()) if self._special_tokens else 0)
        specials = []
        for text, rank in self._special_tokens.items():
            it = _core.VocabItem()
            it.rank = int(rank)
            it.token_bytes = list(text.encode("utf-8"))
            it.token_string = text
            specials.append(it)
        try:
            self._core_bpe = _core.CoreBPE(self.pattern, _vocab_items(vocab), specials, device)
        except Exception as e:
            raise TokenDaggerError(f"Failed to initialize CoreBPE: {e}")

    @classmethod
    def from_files(cls, name: str, *, pat_str: str | None = None, tiktoken_model: str | Path | None = None,
                   hf_config: str | Path | None = None, specials_mergeable: bool = False,
                   tekken: str | Path | None = None, vocab_file: str | Path | None = None,
                   special_tokens_file: str | Path | None = None, device: int = -1) -> "Tokenizer":
        """Build a tokenizer straight from vocabulary files with the C++ loaders (no per-token Python objects):
        a tiktoken ``.model`` (+ optional Hugging Face ``tokenizer_config.json`` for the special tokens, which
        ``specials_mergeable`` also enters as ordinary tokens the way the reference's benchmarks do), a Mistral
        ``tekken.json`` (carries its own pattern), or the reference wrapper's JSON files."""
        for p in (tiktoken_model, hf_config, tekken, vocab_file, special_tokens_file):
            if p is not None and not Path(p).exists():
                raise FileNotFoundError(f"Vocabulary file not found: {p}")
        s = lambda p: "" if p is None else str(p)
        self = cls.__new__(cls)
        self.name = name
        try:
            self._core_bpe = _core.CoreBPE.from_files(pat_str or "", s(tiktoken_model), s(hf_config), specials_mergeable,
                                                      s(tekken), s(vocab_file), s(special_tokens_file), device)
        except Exception as e:
            raise TokenDaggerError(f"Failed to initialize CoreBPE: {e}")
        self.pattern = self._core_bpe.pattern()
        self._special_tokens = dict(self._core_bpe.special_map())
        self.max_token_value = int(self._core_bpe.info(3))
        return self

    @staticmethod
    def _read_json(path, what):
        p = Path(path)
        if not p.exists():
            raise FileNotFoundError(f"{what} file not found: {p}")
        with open(p, "r", encoding="utf-8") as f:
            return json.load(f)

    def __repr__(self) -> str:
        return f"<TokenDagger {self.name!r}>"

    # ------------------------------------------------------------------ encoding ---------------
    def _special_sets(self, allowed_special, disallowed_special):
        if allowed_special == "all":
            allowed_special = set(self._special_tokens)
        if disallowed_special == "all":
            disallowed_special = set(self._special_tokens) - set(allowed_special)
        return set(allowed_special), disallowed_special

    @staticmethod
    def _check_disallowed(text: str, disallowed_special):
        for token in disallowed_special or ():
            if token in text:
                raise ValueError(f"Encountered disallowed special token {token!r}. "
                                 f"Pass it to allowed_special to encode it as a special token.")

    def encode_ordinary(self, text: str) -> list[int]:
        try:
            return self._core_bpe.encode_ordinary(text)
        except Exception as e:
            raise TokenDaggerError(f"Encoding failed: {e}")

    def encode(
        self,
        text: str,
        *,
        allowed_special: Literal["all"] | AbstractSet[str] = set(),
        disallowed_special: Literal["all"] | Collection[str] = set(),
    ) -> list[int]:
        allowed, disallowed = self._special_sets(allowed_special, disallowed_special)
        self._check_disallowed(text, disallowed)
        try:
            tokens, _ = self._core_bpe.encode(text, allowed)
            return tokens
        except Exception as e:
            raise TokenDaggerError(f"Encoding failed: {e}")

    def encode_with_special_tokens(self, text: str) -> list[int]:
        try:
            return self._core_bpe.encode_with_special_tokens(text)
        except Exception as e:
            raise TokenDaggerError(f"Encoding failed: {e}")

    def encode_batch(
        self,
        text: Sequence[str],
        *,
        num_threads: int = 8,
        allowed_special: Literal["all"] | AbstractSet[str] = set(),
        disallowed_special: Literal["all"] | Collection[str] = set(),
    ) -> list[list[int]]:
        allowed, disallowed = self._special_sets(allowed_special, disallowed_special)
        texts = text if isinstance(text, (list, tuple)) else list(text)  # (the binding takes a private tuple of the items itself)
        if disallowed:  # (nothing to look for otherwise: the loop alone was 0.2 us per document)
            for t in texts:
                self._check_disallowed(t, disallowed)
        try:
            if allowed:  # every text is cut at its allowed special tokens on the host; all ordinary segments of all
                return self._core_bpe.encode_batch_special(texts, allowed)  # texts run on the GPU as one batch
            return self._core_bpe.encode_batch(texts, MODE_ENCODE)
        except Exception as e:
            raise TokenDaggerError(f"Encoding failed: {e}")

    def encode_ordinary_batch(self, text: Sequence[str], *, num_threads: int = 8) -> list[list[int]]:
        try:
            return self._core_bpe.encode_batch(list(text), MODE_ORDINARY)
        except Exception as e:
            raise TokenDaggerError(f"Encoding failed: {e}")

    def encode_to_numpy(self, text: str | bytes) -> np.ndarray:
        """tiktoken's array-returning encode: int32 ids without a Python int per token."""
        data = text.encode("utf-8") if isinstance(text, str) else bytes(text)
        buf = np.frombuffer(data, dtype=np.uint8)
        try:
            toks, _ = self._core_bpe.encode_batch_numpy(buf, np.asarray([0, len(buf)], dtype=np.int64), MODE_ENCODE)
            return toks
        except Exception as e:
            raise TokenDaggerError(f"Encoding failed: {e}")

    def encode_batch_to_numpy(self, text: np.ndarray | bytes, offsets: np.ndarray, *, ordinary: bool = False):
        """Bulk form: concatenated UTF-8 bytes + int64 document offsets -> (int32 ids, int64 token offsets)."""
        buf = np.frombuffer(text, dtype=np.uint8) if isinstance(text, (bytes, bytearray)) else text
        try:
            return self._core_bpe.encode_batch_numpy(buf, np.asarray(offsets, dtype=np.int64),
                                                     MODE_ORDINARY if ordinary else MODE_ENCODE)
        except Exception as e:
            raise TokenDaggerError(f"Encoding failed: {e}")

    # ------------------------------------------------------------------ decoding ---------------
    def decode_bytes(self, tokens: Sequence[int]) -> bytes:
        try:
            return self._core_bpe.decode_to_bytes(np.asarray(list(tokens), dtype=np.int32))
        except Exception as e:
            raise TokenDaggerError(f"Decoding failed: {e}")

    def decode(self, tokens: Sequence[int], errors: str = "replace") -> str:
        try:
            return self.decode_bytes(tokens).decode("utf-8", errors=errors)
        except TokenDaggerError:
            raise
        except Exception as e:
            raise TokenDaggerError(f"Decoding failed: {e}")

    def decode_batch(self, tokens: Sequence[Sequence[int]], *, num_threads: int = 8, errors: str = "replace") -> list[str]:
        """All documents in one device pass (reference: a thread pool of decode calls, wrapper.py:237-256)."""
        try:
            return [b.decode("utf-8", errors=errors) for b in self._core_bpe.decode_batch([list(t) for t in tokens])]
        except Exception as e:
            raise TokenDaggerError(f"Decoding failed: {e}")

    def decode_bytes_batch(self, tokens: Sequence[Sequence[int]]) -> list[bytes]:
        try:
            return self._core_bpe.decode_batch([list(t) for t in tokens])
        except Exception as e:
            raise TokenDaggerError(f"Decoding failed: {e}")

    def decode_single_token_bytes(self, token: int) -> bytes:
        """tiktoken semantics: KeyError for an id that is not in the vocabulary (host table lookup, no launch)."""
        b = self._core_bpe.token_bytes(int(token))
        if b is None:
            raise KeyError(token)
        return b

    def decode_tokens_bytes(self, tokens: Sequence[int]) -> list[bytes]:
        return [self.decode_single_token_bytes(t) for t in tokens]

    def encode_single_token(self, text_or_bytes: str | bytes) -> int:
        """tiktoken semantics: the id of exactly one token (ordinary or special), KeyError otherwise."""
        data = text_or_bytes.encode("utf-8") if isinstance(text_or_bytes, str) else bytes(text_or_bytes)
        tid = self._core_bpe.single_token(data)
        if tid is None:
            raise KeyError(text_or_bytes)
        return tid

    def token_byte_values(self) -> list[bytes]:
        """Byte strings of all ordinary tokens, sorted (tiktoken.Encoding.token_byte_values)."""
        special = set(self._special_tokens.values())
        out = []
        for t in range(self.max_token_value + 1):
            if t in special:
                continue
            b = self._core_bpe.token_bytes(t)
            if b is not None:
                out.append(b)
        return sorted(out)

    @property
    def eot_token(self) -> int:
        return self._special_tokens["<|endoftext|>"]

    # ------------------------------------------------------------------ utilities --------------
    def special_tokens(self) -> list[str]:
        try:
            return self._core_bpe.special_tokens()
        except Exception as e:
            raise TokenDaggerError(f"Failed to get special tokens: {e}")

    @property
    def special_tokens_set(self) -> set[str]:
        return set(self._special_tokens)

    @property
    def n_vocab(self) -> int:
        return self.max_token_value + 1

    def is_special_token(self, token: int) -> bool:
        return token in self._special_tokens.values()


def load_tokenizer(name: str, vocab_file: str | Path, pattern: str, special_tokens_file: str | Path | None = None) -> Tokenizer:
    """Reference: wrapper.py:333-355.  The JSON files are read by the C++ loader (td_vocab_load_json)."""
    return Tokenizer.from_files(name, pat_str=pattern, vocab_file=vocab_file, special_tokens_file=special_tokens_file)


def load_tiktoken_bpe(path: str | Path) -> dict[bytes, int]:
    """``tiktoken.load.load_tiktoken_bpe`` for a local ``.model`` / ``.tiktoken`` file (C++ loader)."""
    from . import capi
    return capi.load_tiktoken_bpe(path)


def create_tokenizer(name: str, pattern: str, vocab: list[dict], special_tokens: dict[str, int] | None = None) -> Tokenizer:
    return Tokenizer(name=name, pattern=pattern, vocab=vocab, special_tokens=special_tokens)


def Encoding(name: str, *, pat_str: str, mergeable_ranks: dict[bytes, int],
             special_tokens: dict[str, int] | None = None) -> Tokenizer:
    """tiktoken-compatible factory (reference: wrapper.py:382-395)."""
    return Tokenizer(name=name, pat_str=pat_str, mergeable_ranks=mergeable_ranks, special_tokens=special_tokens or {})


def llama4_scout(device: int = -1) -> Tokenizer:
    """The Llama-4-Scout tokenizer from the bundled TDV1 vocabulary (tokendagger_amd/data)."""
    from . import vocab_io
    name, pat, ranks, special = vocab_io.load_tdv(vocab_io.default_vocab_path())
    return Tokenizer(name, pat_str=pat, mergeable_ranks=ranks, special_tokens=special, device=device)
