"""Copy the summaries of a tools/r3_final.sh run (gpurun_out/<tag>/) into profiles/: kernel stats, HBM traffic, hbm_traffic.json.
usage: python tools/r3_publish.py <tag>"""
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
src = ROOT / "gpurun_out" / sys.argv[1]
prof = ROOT / "profiles"
shutil.copy(src / "stats_english_1024.txt", prof / "r3_01_kernel_stats_english_1024.txt")
with open(prof / "r3_02_kernel_stats_other_corpora.txt", "w") as f:
    for name in ("stats_mixed_256.txt", "stats_code_files_256.txt", "stats_chat_256.txt"):
        f.write((src / name).read_text().rstrip("\n") + "\n\n")
t = json.loads((src / "hbm_traffic.json").read_text())
lines = ["# HBM traffic per launch, rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes of the bench command (tools/r3_measure.sh)",
         "# bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (counter unit KiB; factor 2: the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md), averaged over the launches",
         f"# kernel sources {t['kernel_source_sha']} ({t['source']})", ""]
ALG = {"english_llama4_1024": 1970829960}  # text + 4 x ids + 8 x (documents + 1)
for key in sorted(k for k, v in t.items() if isinstance(v, dict)):
    tot = t[key]["_all"]
    head = f"## {key}: all kernels of a step {tot / 1e9:.3f} GB"
    if key in ALG:
        head += f" = {tot / ALG[key]:.2f} x algorithmic ({ALG[key] / 1e9:.3f} GB)"
    lines.append(head)
    for k, v in sorted(t[key]["per_kernel"].items(), key=lambda kv: -kv[1]):
        lines.append(f"  {k:<32} {v / 1e6:8.1f} MB")
    lines.append("")
(prof / "r3_03_hbm_traffic.txt").write_text("\n".join(lines))
shutil.copy(src / "hbm_traffic.json", prof / "hbm_traffic.json")
print("published", src, "->", prof)
