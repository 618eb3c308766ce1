#!/usr/bin/env python
"""Copy the summaries of the last tools/measure_round.sh + PMC/traffic passes from gpurun_out/ into profiles/.
usage: tools/publish_profiles.py <stats tag, e.g. r1_10> <pmc tag, e.g. r1_11> "<headline describing the kernels>" """
import json, shutil, sys
from pathlib import Path
R = Path(__file__).resolve().parents[1]
O, P = R / "gpurun_out" / "r1_final", R / "profiles"
stats_tag, pmc_tag, head = sys.argv[1], sys.argv[2], sys.argv[3]
(P / f"{stats_tag}_kernel_stats.txt").write_text("\n".join((O / f).read_text() for f in ("stats_summary.txt", "stats_code_summary.txt", "stats_mixed_summary.txt")))
pmc = (R / "gpurun_out" / "r1_final_pmc" / "summary.txt").read_text().splitlines()
tr = (R / "gpurun_out" / "r1_final_traffic" / "summary.txt").read_text()
keep = [l for l in pmc if any(k in l for k in ("split_tiles", "encode_tiles", "pack_tokens"))]
(P / f"{pmc_tag}_pmc_and_traffic.txt").write_text(
    f"# MI355X, {head}: python tools/gpu_ablate.py english 256 0 under rocprofv3 --kernel-trace --pmc ... (tools/gpu_pmc.sh, tools/gpu_traffic.sh, tools/pmc_summary.py)\n"
    "# SQ counters: three separate passes; averages per kernel launch, in millions; SQ_*_CYCLES are quad-cycles\n" + "\n".join(keep) +
    "\n\n# HBM traffic: separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes; counter units = KB; per launch, raw\n"
    "# (FETCH_SIZE under-counts wide coalesced reads 2x on gfx950, MI355X_MICROARCH.md: td_encode_tiles traffic = 2*FETCH + WRITE)\n" + tr)
vals = {}
for l in tr.splitlines():
    f = l.split()
    if len(f) == 3 and "encode_tiles" in f[1]:
        vals[f[0]] = float(f[2].split("=")[1])
tot = int(2 * vals["FETCH_SIZE"] * 1000 + vals["WRITE_SIZE"] * 1000)
(P / "hbm_traffic.json").write_text(json.dumps({"english_256": tot, "_note": f"td_encode_tiles, bytes per launch: 2*FETCH_SIZE + WRITE_SIZE from profiles/{pmc_tag}_pmc_and_traffic.txt (gfx950 FETCH_SIZE correction per MI355X_MICROARCH.md)"}))
(P / "r1_bench").mkdir(exist_ok=True)
for f in ("bench_english_256", "bench_english_1024", "bench_code_256", "bench_mixed_256", "bench_mixed_tekken_256", "bench_dist1"):
    shutil.copy(O / f"{f}.json", P / "r1_bench" / f"{f}.json")
shutil.copy(O / "e2e.txt", P / "r1_bench" / "host_to_host.txt")
shutil.copy(O / "host.txt", P / "r1_bench" / "host.txt")
print("published", stats_tag, pmc_tag, "encode traffic", tot)
