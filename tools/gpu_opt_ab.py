"""A/B of one option on one box, same corpus resident in HBM: python tools/gpu_opt_ab.py <corpus[,corpus..]> <MiB> [steps] [option=v0,v1] [pattern]
Default: TD_OPT_DIRECT = 1,0 (the fused loop places a tile's ids itself / every tile is staged and packed).  Prints the kernel
segments (TD_OPT_PROFILE events) and the whole step of each setting (profile off, graph replay as bench.py runs it), and
whether the settings gave identical ids."""
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
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
optspec = sys.argv[4] if len(sys.argv) > 4 else "DIRECT=1,0"
pattern = sys.argv[5] if len(sys.argv) > 5 else "llama4"
oname, ovals = optspec.split("=")
opt = getattr(capi, "TD_OPT_" + oname)
ovals = [int(v) for v in ovals.split(",")]
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
    results = {}
    for rep in range(2):
        for v in ovals:
            tok.set_option(opt, v)
            tok.set_option(capi.TD_OPT_PROFILE, 1)
            tok.set_option(capi.TD_OPT_GRAPH, 0)
            for _ in range(3):
                tok.encode_device(dt.data_ptr(), n, do.data_ptr(), nd, dk.data_ptr(), cap, dto.data_ptr(), s)
            torch.cuda.synchronize()
            tok.device_status(s)
            tok.profile_read()
            for _ in range(steps):
                tok.encode_device(dt.data_ptr(), n, do.data_ptr(), nd, dk.data_ptr(), cap, dto.data_ptr(), s)
            torch.cuda.synchronize()
            tok.device_status(s)
            sums, k = tok.profile_read_all()
            seg = " + ".join(f"{nm.split('+')[0].replace('td_', '')} {val / k:.3f}" for nm, val in sums.items())
            tok.set_option(capi.TD_OPT_PROFILE, 0)
            tok.set_option(capi.TD_OPT_GRAPH, 1)
            for _ in range(3):
                tok.encode_device(dt.data_ptr(), n, do.data_ptr(), nd, dk.data_ptr(), cap, dto.data_ptr(), s)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                tok.encode_device(dt.data_ptr(), n, do.data_ptr(), nd, dk.data_ptr(), cap, dto.data_ptr(), s)
            torch.cuda.synchronize()
            el = (time.perf_counter() - t0) / steps
            tok.device_status(s)
            toff = dto.cpu().numpy()
            total = int(toff[-1])
            results[v] = (dk[:total].cpu().numpy(), toff)
            print(f"{kind} {n >> 20}MiB {pattern} {oname}={v}: {seg} ms | step {el * 1e3:.3f} ms = {n / el / 1e9:.1f} GB/s | {total} tokens, "
                  f"direct {tok.info(capi.TD_INFO_DIRECT_TILES)}/{(n + 8191) // 8192} (not in time {tok.info(capi.TD_INFO_LB_TIMEOUTS)}) deferred {tok.info(9)} flagged {tok.info(10)} long {tok.info(7)} far {tok.info(8)}", flush=True)
    a, b = results[ovals[0]], results[ovals[-1]]
    print(f"{kind}: identical ids and offsets across the settings: {np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0])}", flush=True)
