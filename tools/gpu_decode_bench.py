"""Device-resident decode rate (ids -> bytes) for DESIGN.md: encode 256 MiB of English, decode the ids back, compare."""
import os, sys, time
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
nd = len(offs) - 1
dt = torch.from_numpy(x).cuda(); do = torch.from_numpy(offs).cuda()
cap = n
dk = torch.empty(cap, dtype=torch.int32, device='cuda'); dto = torch.empty(nd + 1, dtype=torch.int64, device='cuda')
s = torch.cuda.current_stream().cuda_stream
tok.encode_device(dt.data_ptr(), n, do.data_ptr(), nd, dk.data_ptr(), cap, dto.data_ptr(), s)
tok.device_status(s)
ntok = int(dto[nd].item())
out = torch.empty(n + 64, dtype=torch.uint8, device='cuda'); nb = torch.zeros(1, dtype=torch.int64, device='cuda')
for _ in range(3):
    tok.decode_device(dk.data_ptr(), ntok, out.data_ptr(), n + 64, nb.data_ptr(), s)
torch.cuda.synchronize(); tok.device_status(s)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    tok.decode_device(dk.data_ptr(), ntok, out.data_ptr(), n + 64, nb.data_ptr(), s)
e1.record(); torch.cuda.synchronize(); tok.device_status(s)
ms = e0.elapsed_time(e1) / 10
ok = int(nb.item()) == n and bool(torch.equal(out[:n], dt))
print(f"decode {kind} {mb} MiB: {ntok} ids -> {int(nb.item())} bytes in {ms:.3f} ms = {n / ms / 1e6:.1f} GB/s of text "
      f"({(4 * ntok + n) / ms / 1e6:.0f} GB/s ids read + bytes written), round trip {'ok' if ok else 'MISMATCH'}", flush=True)
