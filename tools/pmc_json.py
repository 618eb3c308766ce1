#!/usr/bin/env python
"""profiles/r<N>_0x_pmc_*.txt (tools/pmc_workload.sh + tools/pmc_table.py, copied by tools/publish_round.py) -> profiles/pmc_counters.json:
per workload and kernel the figures bench.py quotes in roofline.counters — share of the chip's SIMD cycles that issued a vector instruction,
lanes active per vector instruction, where the wavefronts' resident cycles went (active / waiting to issue / parked on s_waitcnt or a barrier),
wavefronts resident per SIMD, L2 hit rate.  Carries the sha of the kernel sources (= the one profiles/hbm_traffic.json of the same run carries).
usage: python tools/pmc_json.py <round number>"""
import json, re, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
rn = int(sys.argv[1])
prof = ROOT / "profiles"
sha = json.loads((prof / "hbm_traffic.json").read_text())["kernel_source_sha"]
out = {"kernel_source_sha": sha, "source": f"profiles/r{rn}_04..06_pmc_*.txt (separate rocprofv3 --kernel-trace --pmc passes of the bench command)"}
for f in sorted(prof.glob(f"r{rn}_0[456]_pmc_*.txt")):
    key = re.sub(r"^r\d+_0\d_pmc_", "", f.stem)
    ent = {}
    for line in f.read_text().splitlines():
        m = re.match(r"^(td_\S+(?:<[^>]*>)?)\s+(GRBM_GUI_ACTIVE=.*)$", line)
        if not m:
            continue
        k = m.group(1).split("<")[0]
        d = {a: float(b) for a, b in (kv.split("=") for kv in m.group(2).split())}
        gui, wc = d.get("GRBM_GUI_ACTIVE", 0.0), d.get("SQ_WAVE_CYCLES", 0.0)
        if gui < 8 * 24000 or not wc:  # (kernels of under 10 us: nothing to say)
            continue
        dur = gui / 8
        stalls = {"parked on s_waitcnt / a barrier": d.get("SQ_WAIT_ANY", 0) / wc, "waiting to issue": d.get("SQ_WAIT_INST_ANY", 0) / wc,
                  "issuing": d.get("SQ_ACTIVE_INST_ANY", 0) / wc}
        ent[k] = {"us_at_2.4GHz": round(dur / 2400, 1),
                  "valu_busy_if_4_cycles_per_instruction": round(d.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (dur * 1024), 3),
                  "lanes_per_valu_instruction": round(d.get("SQ_THREAD_CYCLES_VALU", 0) / max(d.get("SQ_ACTIVE_INST_VALU", 1), 1), 1),
                  "wave_cycles": {k2: round(v, 3) for k2, v in stalls.items()},
                  "top_wave_state": max(stalls, key=stalls.get),
                  "waves_per_simd_resident": round(wc * 4 / (dur * 1024), 2),
                  "valu_instructions_per_input_byte_x64": None,
                  "l2_hit_rate": round(d.get("TCC_HIT_sum", 0) / max(d.get("TCC_HIT_sum", 0) + d.get("TCC_MISS_sum", 0), 1), 3),
                  "lds_bank_conflict_frac": round(d.get("SQ_LDS_BANK_CONFLICT", 0) / d["SQ_LDS_IDX_ACTIVE"], 3) if d.get("SQ_LDS_IDX_ACTIVE") else None,
                  "valu_wave_instructions": int(d.get("SQ_INSTS_VALU", 0)),
                  "valu_note": "SQ_ACTIVE_INST_VALU counts instructions; a wave64 instruction issues in 2 or 4 cycles by kind (profiles/r6_00_valu_issue_rate.txt): the busy figure is an upper bound, x 0.8 for the fused loop's mix"}
        ent[k].pop("valu_instructions_per_input_byte_x64")
    out[key] = ent
(prof / "pmc_counters.json").write_text(json.dumps(out, indent=1, sort_keys=True) + "\n")
print("wrote", prof / "pmc_counters.json", {k: list(v) for k, v in out.items() if isinstance(v, dict)})
