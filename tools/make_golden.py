#!/usr/bin/env python
"""Generate tests/golden/llama4_golden.npz from the COMPILED REFERENCE (oracle/_ref/libtdref.so, built from
/root/reference/src/tiktoken/tiktoken.cpp) — run in the build container only; the output is committed so
the GPU box (no /root/reference) can check parity against known answers.

Contents (Llama-4-Scout vocab with specials merged into mergeable_ranks, as the reference's tests build it):
  text, offsets            all inputs concatenated (uint8) + int64 document offsets
  names                    case name per document
  enc, enc_offsets         CoreBPE::encode(doc, {}) ids             (tiktoken.cpp:169-234)
  ord_same                 1 if CoreBPE::encode_ordinary(doc) gives the same ids (always, for this vocab)
  piece_ends, piece_offsets  split_text piece END offsets per document (tiktoken.cpp:70-128)
  decode_ids / decode_bytes  decode_bytes known answers (tiktoken.cpp:236-255)
Inputs: tests/cases.py, the reference fixtures tests/input/{lorem,emoji}.txt (stripped, as
test_tokendagger_vs_tiktoken.py:217-222 does, and raw), seeded fuzz strings (tests/helpers.py), and
slices of the seeded corpora in td_corpus.py.
"""
import random
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import cases  # noqa: E402
import helpers as H  # noqa: E402
import td_corpus  # noqa: E402

REF_INPUT = Path("/root/reference/tests/input")


def main():
    R = H.ref_tokenizer()
    docs, names = [], []
    for name, s in cases.all_strings():
        docs.append(s.encode("utf-8")); names.append(name)
    for f in ("lorem", "emoji"):
        raw = (REF_INPUT / f"{f}.txt").read_bytes()
        docs.append(raw.decode("utf-8").strip().encode("utf-8")); names.append(f"file_{f}_stripped")
        docs.append(raw); names.append(f"file_{f}_raw")
    rng = random.Random(20250925)
    for i in range(1500):
        docs.append(H.fuzz_string(rng).encode("utf-8")); names.append(f"fuzz[{i}]")
    for i in range(1500):
        docs.append(H.random_unicode_string(rng).encode("utf-8")); names.append(f"uni[{i}]")
    for nm, gen in (("english", td_corpus.english), ("mixed", td_corpus.mixed), ("code", td_corpus.code)):
        x, o = gen(192 * 1024, seed=1)
        docs.append(x.tobytes()); names.append(f"corpus_{nm}_192k")  # one big document
        for d in range(min(40, len(o) - 1)):                        # and its first documents one by one
            docs.append(x[o[d]:o[d + 1]].tobytes()); names.append(f"corpus_{nm}_doc[{d}]")
    text, offs = H.pack_docs(docs)
    enc, enc_offs, ord_same, pe, pe_offs = [], [0], [], [], [0]
    for d in docs:
        e = R.encode(d)
        o = R.encode_ordinary(d)
        ord_same.append(int(np.array_equal(e, o)))
        enc.append(e); enc_offs.append(enc_offs[-1] + len(e))
        p = R.split(d) if len(d) else np.zeros(0, np.int64)
        pe.append(p); pe_offs.append(pe_offs[-1] + len(p))
    dec_ids = cases.DECODE_IDS
    dec_bytes = [R.decode_bytes(np.asarray(t, dtype=np.int32)) for t in dec_ids]
    out = ROOT / "tests" / "golden" / "llama4_golden.npz"
    np.savez_compressed(
        out, text=np.frombuffer(text, dtype=np.uint8), offsets=offs, names=np.asarray(names),
        enc=np.concatenate(enc).astype(np.int32), enc_offsets=np.asarray(enc_offs, dtype=np.int64),
        ord_same=np.asarray(ord_same, dtype=np.uint8),
        piece_ends=np.concatenate(pe).astype(np.int64), piece_offsets=np.asarray(pe_offs, dtype=np.int64),
        decode_ids=np.asarray([np.asarray(t, dtype=np.int32) for t in dec_ids], dtype=object),
        decode_bytes=np.asarray(dec_bytes, dtype=object),
        pcre2_version=np.asarray(R.pcre2_version()))
    print(f"wrote {out}: {len(docs)} docs, {len(text)} bytes, {enc_offs[-1]} tokens, ord_same={sum(ord_same)}/{len(docs)}, "
          f"{out.stat().st_size} bytes on disk")


def tekken_style():
    """Second fixture: the SAME input documents through the compiled reference with the Mistral tekken split pattern
    (Llama-4 vocabulary: the tekken.json vocabulary is absent from the reference checkout) -> tekken_style_golden.npz
    {enc, enc_offsets, piece_ends, piece_offsets}; inputs are llama4_golden.npz's text/offsets."""
    g = np.load(ROOT / "tests" / "golden" / "llama4_golden.npz", allow_pickle=True)
    text, offs = g["text"].tobytes(), g["offsets"]
    R = H.ref_tokenizer_tekken()
    enc, enc_offs, pe, pe_offs = [], [0], [], [0]
    for d in range(len(offs) - 1):
        doc = text[offs[d]:offs[d + 1]]
        e = R.encode(doc)
        assert np.array_equal(e, R.encode_ordinary(doc))
        enc.append(e); enc_offs.append(enc_offs[-1] + len(e))
        p = R.split(doc) if len(doc) else np.zeros(0, np.int64)
        pe.append(p); pe_offs.append(pe_offs[-1] + len(p))
    out = ROOT / "tests" / "golden" / "tekken_style_golden.npz"
    np.savez_compressed(out, pattern=np.asarray(H.TEKKEN_PAT), enc=np.concatenate(enc).astype(np.int32),
                        enc_offsets=np.asarray(enc_offs, dtype=np.int64), piece_ends=np.concatenate(pe).astype(np.int64),
                        piece_offsets=np.asarray(pe_offs, dtype=np.int64))
    print(f"wrote {out}: {len(offs) - 1} docs, {enc_offs[-1]} tokens, {out.stat().st_size} bytes on disk")


def cl100k_style():
    """Third fixture: the same documents through the compiled reference with the cl100k_base / Llama-3 split pattern
    (Llama-4 vocabulary) -> cl100k_style_golden.npz."""
    g = np.load(ROOT / "tests" / "golden" / "llama4_golden.npz", allow_pickle=True)
    text, offs = g["text"].tobytes(), g["offsets"]
    R = H.ref_tokenizer_cl100k()
    enc, enc_offs, pe, pe_offs = [], [0], [], [0]
    for d in range(len(offs) - 1):
        doc = text[offs[d]:offs[d + 1]]
        e = R.encode(doc)
        enc.append(e); enc_offs.append(enc_offs[-1] + len(e))
        p = R.split(doc) if len(doc) else np.zeros(0, np.int64)
        pe.append(p); pe_offs.append(pe_offs[-1] + len(p))
    out = ROOT / "tests" / "golden" / "cl100k_style_golden.npz"
    np.savez_compressed(out, pattern=np.asarray(H.CL100K_PAT), enc=np.concatenate(enc).astype(np.int32),
                        enc_offsets=np.asarray(enc_offs, dtype=np.int64), piece_ends=np.concatenate(pe).astype(np.int64),
                        piece_offsets=np.asarray(pe_offs, dtype=np.int64))
    print(f"wrote {out}: {len(offs) - 1} docs, {enc_offs[-1]} tokens, {out.stat().st_size} bytes on disk")


def gpt2_style():
    """Fourth fixture: the same documents through the compiled reference with the GPT-2 split pattern (Llama-4
    vocabulary) -> gpt2_style_golden.npz."""
    g = np.load(ROOT / "tests" / "golden" / "llama4_golden.npz", allow_pickle=True)
    text, offs = g["text"].tobytes(), g["offsets"]
    R = H.ref_tokenizer_gpt2()
    enc, enc_offs, pe, pe_offs = [], [0], [], [0]
    for d in range(len(offs) - 1):
        doc = text[offs[d]:offs[d + 1]]
        e = R.encode(doc)
        enc.append(e); enc_offs.append(enc_offs[-1] + len(e))
        p = R.split(doc) if len(doc) else np.zeros(0, np.int64)
        pe.append(p); pe_offs.append(pe_offs[-1] + len(p))
    out = ROOT / "tests" / "golden" / "gpt2_style_golden.npz"
    np.savez_compressed(out, pattern=np.asarray(H.GPT2_PAT), enc=np.concatenate(enc).astype(np.int32),
                        enc_offsets=np.asarray(enc_offs, dtype=np.int64), piece_ends=np.concatenate(pe).astype(np.int64),
                        piece_offsets=np.asarray(pe_offs, dtype=np.int64))
    print(f"wrote {out}: {len(offs) - 1} docs, {enc_offs[-1]} tokens, {out.stat().st_size} bytes on disk")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "gpt2":
        gpt2_style()
    elif len(sys.argv) > 1 and sys.argv[1] == "cl100k":
        cl100k_style()
    elif len(sys.argv) > 1 and sys.argv[1] == "tekken":
        tekken_style()
    else:
        main()
