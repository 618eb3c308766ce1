"""Single pieces above 1 KiB, one per call: milliseconds per td_encode (host copies included) with the piece on one workgroup of
td_giant_pieces and on all of them (TD_OPT_GIANT_COOP_MIN), ids compared between the two.  GPU box:  python tools/gpu_giant.py"""
import random
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import helpers as H  # noqa: E402
from tokendagger_amd import capi  # noqa: E402

rng = random.Random(77)
L = "abcdefghijklmnopqrstuvwxyz"
docs = [("2 KB random letters", "".join(rng.choice(L) for _ in range(2_000)).encode()),
        ("8 KB random letters", "".join(rng.choice(L) for _ in range(8_000)).encode()),
        ("20 KB random letters", "".join(rng.choice(L) for _ in range(20_000)).encode()),
        ("50 KB random letters", "".join(rng.choice(L) for _ in range(50_000)).encode()),
        ("200 KB random letters", "".join(rng.choice(L) for _ in range(200_000)).encode()),
        ("1 MB random letters", "".join(rng.choice(L) for _ in range(1_000_000)).encode()),
        ("4 MB random letters", "".join(rng.choice(L) for _ in range(3_000_000)).encode()),
        ("400 KB skewed letters", "".join(rng.choice("eeeeeeetttttaaaaooooiiinnnssshhrrdlcumwfgypbvkjxqz") for _ in range(400_000)).encode()),
        ("300 KB DNA", "".join(rng.choice("ACGT") for _ in range(300_000)).encode()),
        ("1 MB of one letter", b"a" * 1_000_000), ("1 MB of blanks", b" " * 1_000_000), ("120 KB abab", ("ab" * 60_000).encode()),
        ("70 KB =", ("=" * 70_000 + "\n").encode()), ("90 KB CJK run", ("的" * 30_000).encode("utf-8")), ("20 KB binary digits", "".join(rng.choice("01") for _ in range(20_000)).encode())]
pat, mr, special = H.llama4()
tok = capi.HipTokenizer(pat, mr, special, device=0)
print(f"{'piece':28s} {'bytes':>9s} {'one workgroup, ms':>18s} {'all workgroups, ms':>19s} {'ids':>8s}")
for name, d in docs:
    res = []
    for limit in (1 << 30, 1024):
        tok.set_option(capi.TD_OPT_GIANT_COOP_MIN, limit)
        if limit > 1024 and len(d) > 1_500_000:
            res.append((float("nan"), None))
            continue
        tok.encode(d)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            ids = tok.encode(d)
            best = min(best, time.perf_counter() - t0)
        res.append((best * 1e3, ids))
    if res[0][1] is not None:
        assert np.array_equal(res[0][1], res[1][1]), name
    print(f"{name:28s} {len(d):9d} {res[0][0]:18.2f} {res[1][0]:19.2f} {len(res[1][1]):8d}")
