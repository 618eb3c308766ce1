"""hipcc -Rpass-analysis=kernel-resource-usage of the device sources -> profiles/r<round>_resource_usage.txt (usage: tools/resource_usage.py <round>)
(VGPRs, spilled VGPRs, scratch bytes per lane, waves/SIMD, LDS bytes per workgroup of every kernel)."""
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench

out = [f"# hipcc --offload-arch=gfx950 -O3 -Rpass-analysis=kernel-resource-usage on these sources (kernel source sha {bench.kernel_source_sha()})",
       "# kernel | VGPRs | spilled VGPRs | scratch B/lane | waves/SIMD | LDS B/workgroup"]
for src in ("td_kernels.hip", "td_generic.hip", "td_special.hip"):
    p = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{ROOT / 'include'}", "-c",
                        str(ROOT / "tokendagger_amd" / "csrc" / src), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"],
                       capture_output=True, text=True)
    cur = {}
    for line in p.stderr.splitlines():
        m = re.search(r"remark: +(Function Name|VGPRs|VGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (.*?) \[-Rpass", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()}
        cur[k] = v
        if k.startswith("LDS"):
            name = cur["name"].replace("td::", "").replace("(td::EncodeArgs)", "")
            out.append(f"{name} | {cur['VGPRs']} | {cur['VGPRs Spill']} | {cur['ScratchSize [bytes/lane]']} | {cur['Occupancy [waves/SIMD]']} | {v}")
(ROOT / "profiles" / f"r{int(sys.argv[1]) if len(sys.argv) > 1 else 6}_resource_usage.txt").write_text("\n".join(out) + "\n")
print("\n".join(out))
