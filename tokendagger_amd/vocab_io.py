"""TDV1 — the on-disk vocabulary container of this repo (own format, little-endian, gzip'd).

    magic   b"TDV1"
    u32     n_vocab, n_special, pat_len, name_len
    bytes   pat_str (UTF-8), name (UTF-8)
    i32[n_vocab]  ranks      u16[n_vocab]  byte lengths      u8[...]  token bytes, concatenated
    i32[n_special] ids       u16[n_special] byte lengths     u8[...]  special-token strings (UTF-8)

It carries the same information the reference loads from a tiktoken ``.model`` file (lines of
``base64 rank``; reference loader: src/main.cpp:89-110) plus the HF ``added_tokens_decoder`` table
(src/main.cpp:121-133) and the split pattern (src/main.cpp:114), so a tokenizer can be rebuilt on a
machine where the reference checkout is absent (the GPU box).
"""
from __future__ import annotations

import base64
import gzip
import json
import struct
from pathlib import Path

import numpy as np

MAGIC = b"TDV1"

LLAMA4_PAT_STR = (
    r"[^\r\n\p{L}\p{N}]?[\p{Lu}\p{Lt}\p{Lm}\p{Lo}\p{M}]*[\p{Ll}\p{Lm}\p{Lo}\p{M}]+(?i:'s|'t|'re|'ve|'m|'ll|'d)?"
    r"|[^\r\n\p{L}\p{N}]?[\p{Lu}\p{Lt}\p{Lm}\p{Lo}\p{M}]+[\p{Ll}\p{Lm}\p{Lo}\p{M}]*(?i:'s|'t|'re|'ve|'m|'ll|'d)?"
    r"|\p{N}{1,3}| ?[^\s\p{L}\p{N}]+[\r\n/]*|\s*[\r\n]+|\s+(?!\S)|\s+"
)

# cl100k_base (GPT-4) / Llama-3 split pattern
CL100K_PAT_STR = (
    r"(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}{1,3}| ?[^\s\p{L}\p{N}]+[\r\n]*|\s*[\r\n]+|\s+(?!\S)|\s+"
)
# tiktoken's later spelling of the same language (possessive quantifiers)
CL100K_PAT_STR_POSSESSIVE = (
    r"'(?i:[sdmt]|ll|ve|re)|[^\r\n\p{L}\p{N}]?+\p{L}+|\p{N}{1,3}| ?[^\s\p{L}\p{N}]++[\r\n]*|\s*[\r\n]|\s+(?!\S)|\s+"
)

# Qwen2 / Qwen2.5 / Qwen3 (tokenizer.json pre_tokenizer Split regex): cl100k_base with single-digit number pieces
QWEN2_PAT_STR = (
    r"(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}| ?[^\s\p{L}\p{N}]+[\r\n]*|\s*[\r\n]+|\s+(?!\S)|\s+"
)
# cl100k_base as current tiktoken releases spell it (tiktoken_ext/openai_public.py): the same language again
CL100K_PAT_STR_CURRENT = (
    r"'(?i:[sdmt]|ll|ve|re)|[^\r\n\p{L}\p{N}]?+\p{L}++|\p{N}{1,3}+| ?[^\s\p{L}\p{N}]++[\r\n]*+|\s++$|\s*[\r\n]|\s+(?!\S)|\s"
)

# GPT-2 (r50k_base / p50k_base) split pattern
GPT2_PAT_STR = r"'s|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+"

GPT2_PAT_STR_POSSESSIVE = r"'(?:[sdmt]|ll|ve|re)| ?\p{L}++| ?\p{N}++| ?[^\s\p{L}\p{N}]++|\s++$|\s+(?!\S)|\s"

# Mistral tekken.json config.pattern (reference loads it from the file: tests/throughput_test.py:118): the Llama-4
# pattern without the contraction suffix and with single-digit number pieces.
TEKKEN_PAT_STR = (
    r"[^\r\n\p{L}\p{N}]?[\p{Lu}\p{Lt}\p{Lm}\p{Lo}\p{M}]*[\p{Ll}\p{Lm}\p{Lo}\p{M}]+"
    r"|[^\r\n\p{L}\p{N}]?[\p{Lu}\p{Lt}\p{Lm}\p{Lo}\p{M}]+[\p{Ll}\p{Lm}\p{Lo}\p{M}]*"
    r"|\p{N}| ?[^\s\p{L}\p{N}]+[\r\n/]*|\s*[\r\n]+|\s+(?!\S)|\s+"
)


def save_tdv(path, name: str, pat_str: str, mergeable_ranks: dict[bytes, int],
             special_tokens: dict[str, int]) -> None:
    items = sorted(mergeable_ranks.items(), key=lambda kv: kv[1])
    sp = sorted(special_tokens.items(), key=lambda kv: kv[1])
    pat = pat_str.encode("utf-8")
    nm = name.encode("utf-8")
    out = [MAGIC, struct.pack("<IIII", len(items), len(sp), len(pat), len(nm)), pat, nm]
    out.append(np.asarray([r for _, r in items], dtype="<i4").tobytes())
    out.append(np.asarray([len(b) for b, _ in items], dtype="<u2").tobytes())
    out.append(b"".join(b for b, _ in items))
    sb = [s.encode("utf-8") for s, _ in sp]
    out.append(np.asarray([r for _, r in sp], dtype="<i4").tobytes())
    out.append(np.asarray([len(b) for b in sb], dtype="<u2").tobytes())
    out.append(b"".join(sb))
    with gzip.GzipFile(path, "wb", compresslevel=9, mtime=0) as f:
        f.write(b"".join(out))


def load_tdv(path) -> tuple[str, str, dict[bytes, int], dict[str, int]]:
    """-> (name, pat_str, mergeable_ranks, special_tokens)"""
    raw = gzip.open(path, "rb").read()
    if raw[:4] != MAGIC:
        raise ValueError(f"{path}: not a TDV1 vocabulary file")
    n_vocab, n_special, pat_len, name_len = struct.unpack_from("<IIII", raw, 4)
    p = 20
    pat = raw[p:p + pat_len].decode("utf-8"); p += pat_len
    name = raw[p:p + name_len].decode("utf-8"); p += name_len

    def section(n, p):
        ranks = np.frombuffer(raw, dtype="<i4", count=n, offset=p); p += 4 * n
        lens = np.frombuffer(raw, dtype="<u2", count=n, offset=p); p += 2 * n
        offs = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(lens, out=offs[1:])
        blob = raw[p:p + int(offs[-1])]; p += int(offs[-1])
        return ranks, offs, blob, p

    ranks, offs, blob, p = section(n_vocab, p)
    mergeable = {blob[offs[i]:offs[i + 1]]: int(ranks[i]) for i in range(n_vocab)}
    sranks, soffs, sblob, p = section(n_special, p)
    special = {sblob[soffs[i]:soffs[i + 1]].decode("utf-8"): int(sranks[i]) for i in range(n_special)}
    return name, pat, mergeable, special


def load_tiktoken_model(path) -> dict[bytes, int]:
    """tiktoken ``.model`` text format: one ``<base64 token> <rank>`` per line."""
    ranks: dict[bytes, int] = {}
    with open(path, "rb") as f:
        for line in f:
            line = line.strip()
            if not line:
                continue
            tok, rank = line.split()
            ranks[base64.b64decode(tok)] = int(rank)
    return ranks


def load_hf_added_tokens(path) -> dict[str, int]:
    """HF ``tokenizer_config.json`` -> {content: id} from ``added_tokens_decoder``."""
    cfg = json.loads(Path(path).read_text(encoding="utf-8"))
    return {v["content"]: int(k) for k, v in cfg.get("added_tokens_decoder", {}).items()}


def default_vocab_path(name: str = "llama4_scout") -> Path:
    return Path(__file__).resolve().parent / "data" / f"{name}.tdv.gz"
