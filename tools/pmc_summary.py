#!/usr/bin/env python
"""rocprofv3 --pmc counter_collection CSVs -> per-kernel averages per launch (text for profiles/).
usage: tools/pmc_summary.py <dir with pass subdirs> [scale]   (values printed in millions unless scale=raw)"""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
raw = len(sys.argv) > 2 and sys.argv[2] == "raw"
for pdir in sorted(glob.glob(os.path.join(root, "*"))):
    if not os.path.isdir(pdir):
        continue
    acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
    for f in glob.glob(os.path.join(pdir, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if k.startswith("void "):  # templated kernels carry their return type
                k = k[5:]
            if not k.startswith("td::"):
                continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
    for k in sorted(acc):
        vals = "  ".join(f"{c}={acc[k][c] / cnt[k][c] / (1 if raw else 1e6):.1f}" for c in sorted(acc[k]))
        print(f"{os.path.basename(pdir):<11} {k:<22} {vals}")
