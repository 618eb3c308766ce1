"""One 8 KiB tile through td_encode_device, 300 times (what bench.py's roofline.fixed_overhead_us times): run under
rocprofv3 --kernel-trace --stats to see what each launch of the step costs when it has next to nothing to do."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import td_corpus
from tokendagger_amd import capi, vocab_io
name, pat, ranks, special = vocab_io.load_tdv(vocab_io.default_vocab_path())
tok = capi.HipTokenizer(pat, ranks, special, device=0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
x, _ = td_corpus.english(1 << 20, seed=3)
d_text = torch.from_numpy(x).cuda()
one = torch.tensor([0, n], dtype=torch.int64, device="cuda")
t_one = torch.empty(n + 1024, dtype=torch.int32, device="cuda")
o_one = torch.empty(2, dtype=torch.int64, device="cuda")
s = torch.cuda.current_stream().cuda_stream
tok.set_option(capi.TD_OPT_GRAPH, int(os.environ.get("TD_BENCH_GRAPH", "1")))
for _ in range(3):
    tok.encode_device(d_text.data_ptr(), n, one.data_ptr(), 1, t_one.data_ptr(), n + 1024, o_one.data_ptr(), s)
    torch.cuda.synchronize(); tok.device_status(s)
for _ in range(20):
    tok.encode_device(d_text.data_ptr(), n, one.data_ptr(), 1, t_one.data_ptr(), n + 1024, o_one.data_ptr(), s)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300):
    tok.encode_device(d_text.data_ptr(), n, one.data_ptr(), 1, t_one.data_ptr(), n + 1024, o_one.data_ptr(), s)
torch.cuda.synchronize()
print(f"{n} bytes: {(time.perf_counter() - t0) / 300 * 1e6:.1f} us per step, sequence {'sparse' if tok.info(capi.TD_INFO_SPARSE) == 1 else 'dense'}")
