"""Do a pinned H2D copy and a pinned D2H copy on two streams run side by side on this box?  (td_encode_batch's pipeline assumes so.)
GPU box:  python tools/gpu_pcie_duplex.py   (HSA_ENABLE_SDMA=0 in the environment: the copies as blit kernels)"""
import os, statistics, time
import torch
dev = torch.device("cuda:0")
n_in, n_out = 1 << 30, 885 << 20
hp = torch.empty(n_in, dtype=torch.uint8).pin_memory(); hp.fill_(7)
dp = torch.empty(n_in, dtype=torch.uint8, device=dev)
ho = torch.empty(n_out, dtype=torch.uint8).pin_memory()
do = torch.zeros(n_out, dtype=torch.uint8, device=dev)
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
def timed(f, reps=3):
    f(); torch.cuda.synchronize(dev)
    rs = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(dev); rs.append(time.perf_counter() - t0)
    return statistics.median(rs)
def h2d(chunks=1):
    with torch.cuda.stream(s1):
        k = n_in // chunks
        for c in range(chunks): dp[c * k:(c + 1) * k].copy_(hp[c * k:(c + 1) * k], non_blocking=True)
def d2h(chunks=1):
    with torch.cuda.stream(s2):
        k = n_out // chunks
        for c in range(chunks): ho[c * k:(c + 1) * k].copy_(do[c * k:(c + 1) * k], non_blocking=True)
print("HSA_ENABLE_SDMA =", os.environ.get("HSA_ENABLE_SDMA"))
for chunks in (1, 32):
    a, b = timed(lambda: h2d(chunks)), timed(lambda: d2h(chunks))
    c = timed(lambda: (h2d(chunks), d2h(chunks)))
    print(f"chunks {chunks:3d}: H2D 1 GiB {a * 1e3:6.2f} ms ({n_in / a / 1e9:5.1f} GB/s) | D2H 885 MiB {b * 1e3:6.2f} ms ({n_out / b / 1e9:5.1f} GB/s) | both, two streams {c * 1e3:6.2f} ms "
          f"(sum {1e3 * (a + b):.2f}, max {1e3 * max(a, b):.2f})", flush=True)
