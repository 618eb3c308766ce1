#!/usr/bin/env python
"""The lines of a closing run (tools/final_round.sh <tag>) that DESIGN.md / README.md quote.  usage: tools/round_summary.py <tag>"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out", sys.argv[1])
def last_json(p):
    return json.loads(open(p).read().strip().splitlines()[-1])
log = open(os.path.join(O, "pytest_gpu.log")).read()
print("pytest -m gpu:", (re.findall(r"\d+ passed[^\n]*|\d+ failed[^\n]*", log) or ["?"])[-1])
print(open(os.path.join(O, "smoke.log")).read().strip().splitlines()[-1])
j = last_json(os.path.join(O, "bench_default.json")); r = j["roofline"]
print(f'headline {j["value"]:.1f} GB/s  {j["ms_per_step"]:.4f} ms  dominant {r["kernel"]} {r["kernel_ms_avg"]} ms frac {r["frac"]}  whole-step frac {r["whole_step"]["frac"]}  '
      f'fixed {r.get("fixed_overhead_us")} us  verified {j["config"]["verified_vs_oracle"]}  traffic {r.get("traffic")}')
print("launch_mode", {k: v for k, v in j["launch_mode"].items() if k != "launch_sequence_note"})
print("per_kernel", r.get("per_kernel"))
cb = j.get("cpu_baseline", {})
print("cpu_baseline", {k: cb.get(k) for k in ("value", "unit", "cores", "kind")}, "| python", (j.get("drop_in_python") or {}).get("threads_8"), "x", (j.get("drop_in_python") or {}).get("vs_reference_encode_batch"))
print("e2e", j.get("e2e"))
for k, v in (j.get("configs") or {}).items():
    print(f'  {k:<34} {v.get("value")} GB/s  {v.get("ms_per_step")} ms  {v.get("verified_vs_oracle", v.get("error"))}  {v.get("launch_sequence")}  x{round(v.get("value", 0) / j["value"], 3)} of the headline')
try:
    t = last_json(os.path.join(O, "bench_2rank_same_gpu_gloo.json")); print("2 ranks on one GPU (gloo):", t["value"], "GB/s", t["config"]["verified_vs_oracle"])
except Exception as e:  # noqa: BLE001
    print("2-rank line:", e)
for f in ("sequences_ab.txt", "direct_ab.txt"):
    p = os.path.join(O, f)
    if os.path.exists(p):
        print("--", f); print("".join(l for l in open(p) if "amdgpu" not in l).strip()[-1500:])
