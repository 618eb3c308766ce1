#!/usr/bin/env python
"""rocprofv3 --pmc passes (tools/pmc_workload.sh) -> one table per kernel for profiles/: instructions per wavefront,
where the wavefronts' cycles go (active / waiting for an instruction to issue / parked on s_waitcnt or a barrier), VALU
utilisation of the chip, lanes active per VALU instruction, LDS conflicts, L2 hit rate.
usage: tools/pmc_table.py <dir with pass subdirs> "<command>"    SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles."""
import csv, glob, os, sys
from collections import defaultdict
root, title = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int)); launches = defaultdict(int)
for pdir in sorted(glob.glob(os.path.join(root, "*"))):
    if not os.path.isdir(pdir):
        continue
    for f in glob.glob(os.path.join(pdir, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if k.startswith("void "):
                k = k[5:]
            if not k.startswith("td::"):
                continue
            k = k[4:]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
A = {k: {c: acc[k][c] / cnt[k][c] for c in acc[k]} for k in acc}
SIMDS = 256 * 4
print(f"# {title}\n# per kernel LAUNCH (averages over the launches of the run); separate rocprofv3 --kernel-trace --pmc passes (tools/pmc_workload.sh)")
print("# rocprofv3 sums a counter over the 8 XCDs: GRBM_GUI_ACTIVE / 8 = the kernel's duration in cycles (x 1/2.4 GHz = the kernel-trace duration);")
print("# cyc/wave = SQ_WAVE_CYCLES x 4 / waves (SQ_*_CYCLES and SQ_WAIT_* count quad-cycles); VALU busy% = SQ_ACTIVE_INST_VALU x 4 / (duration x 1024 SIMDs):")
print("# SQ_ACTIVE_INST_VALU counts INSTRUCTIONS (= SQ_INSTS_VALU) and a wave64 instruction issues in 2 or 4 cycles by kind (profiles/r6_00_valu_issue_rate.txt), so")
print("# this is the busy share if every instruction took 4 cycles — an upper bound (the fused loop's mix averages ~3.3: x 0.8); lanes/VALU = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU (of 64);")
print("# active / issue-stall / parked = SQ_ACTIVE_INST_ANY / SQ_WAIT_INST_ANY / SQ_WAIT_ANY as shares of the wavefronts' resident cycles (parked = s_waitcnt or barrier)")
hdr = f"{'kernel':<34} {'waves':>9} {'VALU/wave':>10} {'SALU/wave':>10} {'LDS/wave':>9} {'VMEM/wave':>10} {'cyc/wave':>10} {'active%':>8} {'issue-stall%':>12} {'parked%':>8} {'VALU busy%':>10} {'lanes/VALU':>10} {'LDS confl%':>10} {'occ waves/SIMD':>14} {'L2 hit%':>8} {'us @2.4GHz':>10}"
print(hdr)
def g(d, k, default=0.0):
    return d.get(k, default)
for k in sorted(A, key=lambda k: -g(A[k], "GRBM_GUI_ACTIVE")):
    d = A[k]
    w = g(d, "SQ_WAVES"); wc = g(d, "SQ_WAVE_CYCLES")
    if w < 1:
        continue
    gui = g(d, "GRBM_GUI_ACTIVE")
    valu = g(d, "SQ_INSTS_VALU")
    vmem = g(d, "SQ_INSTS_VMEM_RD") + g(d, "SQ_INSTS_VMEM_WR") + g(d, "SQ_INSTS_FLAT")
    pct = lambda x, y: f"{100.0 * x / y:.1f}" if y else "-"
    lanes = f"{g(d, 'SQ_THREAD_CYCLES_VALU') / g(d, 'SQ_ACTIVE_INST_VALU'):.1f}" if g(d, "SQ_ACTIVE_INST_VALU") else "-"
    occ = f"{wc * 4 / (gui / 8) / SIMDS:.2f}" if gui else "-"
    l2 = pct(g(d, "TCC_HIT_sum"), g(d, "TCC_HIT_sum") + g(d, "TCC_MISS_sum"))
    print(f"{k[:34]:<34} {w:>9.0f} {valu / w:>10.1f} {g(d, 'SQ_INSTS_SALU') / w:>10.1f} {g(d, 'SQ_INSTS_LDS') / w:>9.1f} {vmem / w:>10.1f} {wc * 4 / w:>10.0f} "
          f"{pct(g(d, 'SQ_ACTIVE_INST_ANY'), wc):>8} {pct(g(d, 'SQ_WAIT_INST_ANY'), wc):>12} {pct(g(d, 'SQ_WAIT_ANY'), wc):>8} "
          f"{pct(g(d, 'SQ_ACTIVE_INST_VALU') * 4, gui / 8 * SIMDS):>10} {lanes:>10} {pct(g(d, 'SQ_LDS_BANK_CONFLICT'), g(d, 'SQ_LDS_IDX_ACTIVE')):>10} {occ:>14} {l2:>8} {gui / 8 / 2400:>9.1f}")
print("\n# raw averages per launch")
for k in sorted(A):
    print(f"{k:<34} " + "  ".join(f"{c}={A[k][c]:.4g}" for c in sorted(A[k])))
