#!/usr/bin/env python
"""Convert a tiktoken .model + HF tokenizer_config.json into this repo's TDV1 container.

Run once in the build container (where /root/reference exists); the output is committed so the GPU
box, which has no reference checkout, can construct the Llama-4-Scout tokenizer:

    python tools/import_vocab.py --model /root/reference/src/tokenizer.model \
        --config /root/reference/src/tokenizer_config.json --name llama4_scout
"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from tokendagger_amd import vocab_io  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", required=True)
ap.add_argument("--config", required=True)
ap.add_argument("--name", default="llama4_scout")
ap.add_argument("--out", default=None)
a = ap.parse_args()
ranks = vocab_io.load_tiktoken_model(a.model)
special = vocab_io.load_hf_added_tokens(a.config)
out = a.out or vocab_io.default_vocab_path(a.name)
vocab_io.save_tdv(out, a.name, vocab_io.LLAMA4_PAT_STR, ranks, special)
name, pat, r2, s2 = vocab_io.load_tdv(out)
assert r2 == ranks and s2 == special and pat == vocab_io.LLAMA4_PAT_STR
print(f"wrote {out}: {len(ranks)} ranks, {len(special)} specials, {Path(out).stat().st_size} bytes")
