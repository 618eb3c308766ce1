"""A/B on one box: the fused tile loop against the two-kernel form (TD_OPT_FUSED 1 / 0), same corpus resident in HBM.
usage: python tools/gpu_ab.py <corpus[,corpus..]> <MiB> [steps] [pattern]
Prints per-segment kernel times (TD_OPT_PROFILE events), the whole step, and whether the two forms gave identical ids."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import bench
from tokendagger_amd import capi, vocab_io

kinds = (sys.argv[1] if len(sys.argv) > 1 else "english").split(",")
mb = int(sys.argv[2]) if len(sys.argv) > 2 else 256
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
pattern = sys.argv[4] if len(sys.argv) > 4 else "llama4"
modes = [int(v) for v in os.environ.get("TD_AB_MODES", "1,0").split(",")]
name, pat, ranks, special = vocab_io.load_tdv(vocab_io.default_vocab_path())
if pattern == "tekken":
    pat = vocab_io.TEKKEN_PAT_STR
tok = capi.HipTokenizer(pat, ranks, special, device=0)
s = torch.cuda.current_stream().cuda_stream
for kind in kinds:
    x, offs = bench.build_corpus(kind, mb << 20, 1000)
    n, nd = len(x), len(offs) - 1
    dt = torch.from_numpy(x).cuda()
    do = torch.from_numpy(offs).cuda()
    cap = n // 2 + 1024 if kind == "english" else n + 1024
    dk = torch.empty(cap, dtype=torch.int32, device="cuda")
    dto = torch.empty(nd + 1, dtype=torch.int64, device="cuda")
    tok.reserve(n, nd + 1)
    tok.set_option(capi.TD_OPT_PROFILE, 1)
    results = {}
    for fused in modes:
        tok.set_option(capi.TD_OPT_FUSED, fused)
        for _ in range(2):
            tok.encode_device(dt.data_ptr(), n, do.data_ptr(), nd, dk.data_ptr(), cap, dto.data_ptr(), s)
        torch.cuda.synchronize()
        tok.device_status(s)
        tok.profile_read()
        t0 = time.perf_counter()
        for _ in range(steps):
            tok.encode_device(dt.data_ptr(), n, do.data_ptr(), nd, dk.data_ptr(), cap, dto.data_ptr(), s)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / steps
        tok.device_status(s)
        sums, k = tok.profile_read_all()
        seg = " + ".join(f"{nm.split('+')[0].replace('td_', '')} {v / k:.3f}" for nm, v in sums.items())
        toff = dto.cpu().numpy()
        total = int(toff[-1])
        results[fused] = (dk[:total].cpu().numpy(), toff)
        print(f"{kind} {n >> 20}MiB {pattern} fused={fused}: {seg} ms | step {el * 1e3:.3f} ms = {n / el / 1e9:.1f} GB/s | {total} tokens, "
              f"deferred {tok.info(9)} flagged {tok.info(10)} long {tok.info(7)} far {tok.info(8)}", flush=True)
    if len(results) == 2:
        a, b = results[modes[0]], results[modes[1]]
        same = np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0])
        print(f"{kind}: fused and two-kernel ids {'IDENTICAL' if same else 'DIFFER'}", flush=True)
        if not same:
            bad = np.flatnonzero(a[1] != b[1])
            print("  first differing document offset index:", bad[:5], "tokens", len(a[0]), len(b[0]))
            m = min(len(a[0]), len(b[0]))
            bt = np.flatnonzero(a[0][:m] != b[0][:m])
            print("  first differing token index:", bt[:5])
    del dt, do, dk, dto
    torch.cuda.empty_cache()
