"""Copy the summaries of a tools/final_round.sh run (gpurun_out/<tag>/) into profiles/ under the round's names.
usage: python tools/publish_round.py <tag> <round number>"""
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
src = ROOT / "gpurun_out" / sys.argv[1]
rn = int(sys.argv[2])
prof = ROOT / "profiles"
shutil.copy(src / "stats_english_1024.txt", prof / f"r{rn}_01_kernel_stats_english_1024.txt")
with open(prof / f"r{rn}_02_kernel_stats_other_corpora.txt", "w") as f:
    for name in ("stats_mixed_256.txt", "stats_code_files_256.txt", "stats_mixed_tekken_256.txt"):
        if (src / name).exists():
            f.write((src / name).read_text().rstrip("\n") + "\n\n")
t = json.loads((src / "hbm_traffic.json").read_text())
lines = ["# fabric traffic of the L2s per launch, rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes of the bench command (tools/measure_workload.sh)",
         "# bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (counter unit KiB; factor 2: the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md), averaged over the launches",
         f"# kernel sources {t['kernel_source_sha']} ({t['source']})", ""]
ALG = {}  # text + 4 x ids + 8 x (documents + 1), of the bench line's own corpus
try:
    bj = json.loads((src / "bench_default.json").read_text().strip().splitlines()[-1])
    ALG["english_llama4_1024"] = int(bj["roofline"]["algorithmic_bytes_per_launch"])
except Exception:  # noqa: BLE001
    pass
for key in sorted(k for k, v in t.items() if isinstance(v, dict)):
    tot = t[key]["_all"]
    head = f"## {key}: all kernels of a step {tot / 1e9:.3f} GB"
    alg = t[key].get("algorithmic_bytes") or ALG.get(key)
    if alg:
        head += f" = {tot / alg:.2f} x algorithmic ({alg / 1e9:.3f} GB = text + 4 x ids + 8 x (documents + 1), SURVEY 8d)"
    lines.append(head)
    for k, v in sorted(t[key]["per_kernel"].items(), key=lambda kv: -kv[1]):
        lines.append(f"  {k:<32} {v / 1e6:8.1f} MB" + (f"   {v / alg:5.2f} x algorithmic" if alg else ""))
    lines.append("")
(prof / f"r{rn}_03_hbm_traffic.txt").write_text("\n".join(lines))
shutil.copy(src / "hbm_traffic.json", prof / "hbm_traffic.json")
for i, name in enumerate(("pmc_english_1024.txt", "pmc_mixed_256.txt", "pmc_code_files_256.txt")):
    if (src / name).exists():
        shutil.copy(src / name, prof / f"r{rn}_0{4 + i}_{name}")
bdir = prof / f"r{rn}_bench"
bdir.mkdir(exist_ok=True)
for name in ("bench_default.json", "bench_default.time", "latency_files.txt", "resource_usage.txt", "direct_ab.txt", "sequences_ab.txt", "bench_2rank_same_gpu_gloo.json", "pybatch.txt", "host.txt", "pytest_gpu.log", "smoke.log", "latency.txt", "giant_pieces.txt", "hostpath.txt", "bench_128_weak.json"):
    if (src / name).exists():
        shutil.copy(src / name, bdir / name)
print("published", src, "->", prof)
