"""Kernel tuning aid: time td_encode_tiles with the tile loop cut after phase N (results are garbage for N != 0)."""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from tokendagger_amd import capi, vocab_io
import bench
kind = sys.argv[1] if len(sys.argv) > 1 else 'english'
mb = int(sys.argv[2]) if len(sys.argv) > 2 else 256
name, pat, ranks, special = vocab_io.load_tdv(vocab_io.default_vocab_path())
tok = capi.HipTokenizer(pat, ranks, special, device=0)
n = mb << 20
x, offs = bench.build_corpus(kind, n, 1000)
n = len(x)  # (the file set is tiled whole: shorter than asked for)
nd = len(offs) - 1
dt = torch.from_numpy(x).cuda(); do = torch.from_numpy(offs).cuda()
cap = n // 2 + 1024 if kind == 'english' else n
dk = torch.empty(cap, dtype=torch.int32, device='cuda'); dto = torch.empty(nd + 1, dtype=torch.int64, device='cuda')
tok.reserve(n, nd + 1); tok.set_option(capi.TD_OPT_PROFILE, 1)
if os.environ.get('TD_MERGE_MODE'): tok.set_option(98, int(os.environ['TD_MERGE_MODE']))
s = torch.cuda.current_stream().cuda_stream
stops = [int(v) for v in sys.argv[3].split(',')] if len(sys.argv) > 3 else [2, 3, 4, 0]
for stop in stops:
    tok.set_option(99, stop)
    for _ in range(2):
        tok.encode_device(dt.data_ptr(), n, do.data_ptr(), nd, dk.data_ptr(), cap, dto.data_ptr(), s)
    torch.cuda.synchronize(); tok.profile_read()
    t0 = time.perf_counter()
    for _ in range(5):
        tok.encode_device(dt.data_ptr(), n, do.data_ptr(), nd, dk.data_ptr(), cap, dto.data_ptr(), s)
    torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 5
    if hasattr(tok._lib, "td_profile_read_ex"):
        sums, k = tok.profile_read_all()
        seg = " + ".join(f"{nm.split('+')[0].replace('td_', '')} {v / k:.3f}" for nm, v in sums.items())
    else:
        ms0, ms1, k = tok.profile_read()
        seg = f"split {ms0/k:.3f} + encode {ms1/k:.3f}"
    print(f"{kind} {mb}MiB stop_after={stop}: {seg} ms, whole step {el*1e3:.3f} ms, {n/el/1e9:.1f} GB/s", flush=True)
try:
    tok.device_status(s)
except Exception as e:
    print('status (expected garbage for ablations):', e)
