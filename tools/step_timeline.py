#!/usr/bin/env python
"""A rocprofv3 --kernel-trace CSV of `python bench.py ...` -> the timeline of a step: every launch's duration and the gap between the
end of the previous launch and its start (the queue's drain + dispatch + ramp-up), averaged over the last steps of the run.
usage: tools/step_timeline.py <dir with *kernel_trace.csv> [steps to average = 8]"""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]; last = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rows = []
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").replace("td::", "").split("(")[0]
        if k.startswith("td_"):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k, int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)))
rows.sort()
steps, cur = [], []
for r in rows:  # a step starts at td_prepare_mark / td_prepare
    if r[2].startswith("td_prepare") and cur:
        steps.append(cur); cur = []
    cur.append(r)
if cur: steps.append(cur)
sig = tuple(r[2] for r in steps[-1])
use = [s for s in steps if tuple(r[2] for r in s) == sig][-last:]
print(f"# {len(steps)} steps in the trace, the last {len(use)} with the sequence of the final one averaged; times in us")
print(f"{'launch':<34}{'workgroups':>11}{'gap before':>12}{'duration':>10}{'ends at':>10}")
tot_gap = tot_dur = 0.0
for i, name in enumerate(sig):
    gap = sum((s[i][0] - (s[i - 1][1] if i else s[i][0])) for s in use) / len(use) / 1e3
    dur = sum((s[i][1] - s[i][0]) for s in use) / len(use) / 1e3
    end = sum((s[i][1] - s[0][0]) for s in use) / len(use) / 1e3
    tot_gap += gap; tot_dur += dur
    print(f"{name:<34}{use[-1][i][3]:>11}{gap:>12.1f}{dur:>10.1f}{end:>10.1f}")
span = sum(s[-1][1] - s[0][0] for s in use) / len(use) / 1e3
per = sum(b[0][0] - a[0][0] for a, b in zip(use, use[1:])) / max(len(use) - 1, 1) / 1e3
print(f"sum of durations {tot_dur:.1f}, sum of gaps {tot_gap:.1f} (negative = launches overlapped), first start to last end {span:.1f}, step to step {per:.1f}")
