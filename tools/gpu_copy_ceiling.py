"""What a copy can reach on this box (the ceiling td_pack_tokens is held against): a dense device-to-device copy, and the pack's own
pattern — 840 ids of every 4160-slot stage region to a dense output — through torch's strided copy.  usage: python tools/gpu_copy_ceiling.py"""
import time
import torch
def rate(fn, nbytes, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / it
    return nbytes / dt / 1e12, dt * 1e3
n_tiles, stage, ids = 262144, 4160, 840
src = torch.empty(n_tiles * ids, dtype=torch.int32, device="cuda").random_()
dst = torch.empty_like(src)
r, ms = rate(lambda: dst.copy_(src), 2 * src.numel() * 4)
print(f"dense copy of {src.numel() * 4 / 1e6:.0f} MB: {ms:.3f} ms = {r:.2f} TB/s (read + write)")
big = torch.empty(n_tiles, stage, dtype=torch.int32, device="cuda")
out = torch.empty(n_tiles, ids, dtype=torch.int32, device="cuda")
r, ms = rate(lambda: out.copy_(big[:, :ids]), 2 * out.numel() * 4)
print(f"{ids} of every {stage} slots -> dense ({out.numel() * 4 / 1e6:.0f} MB): {ms:.3f} ms = {r:.2f} TB/s (read + write)")
r, ms = rate(lambda: big[:, :ids].copy_(out), 2 * out.numel() * 4)
print(f"dense -> {ids} of every {stage} slots: {ms:.3f} ms = {r:.2f} TB/s (read + write)")
