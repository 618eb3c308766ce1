"""Kernel tuning aid: ONE encode of a corpus (for builds that print from the device)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tokendagger_amd import capi, vocab_io
import bench
kind = sys.argv[1]; mb = int(sys.argv[2])
name, pat, ranks, special = vocab_io.load_tdv(vocab_io.default_vocab_path())
tok = capi.HipTokenizer(pat, ranks, special, device=0)
n = mb << 20
x, offs = bench.build_corpus(kind, n, 1000)
n = len(x)  # (the file set is tiled whole: shorter than asked for)
nd = len(offs) - 1
dt = torch.from_numpy(x).cuda(); do = torch.from_numpy(offs).cuda()
dk = torch.empty(n, dtype=torch.int32, device='cuda'); dto = torch.empty(nd + 1, dtype=torch.int64, device='cuda')
s = torch.cuda.current_stream().cuda_stream
tok.encode_device(dt.data_ptr(), n, do.data_ptr(), nd, dk.data_ptr(), n, dto.data_ptr(), s)
torch.cuda.synchronize()
tok.device_status(s)
print("long pieces", tok.info(capi.TD_INFO_LONG_PIECES), "far pieces", tok.info(capi.TD_INFO_FAR_PIECES))
