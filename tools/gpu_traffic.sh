#!/usr/bin/env bash
# HBM traffic of the tile kernels: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (TCC slot limits;
# MI355X_MICROARCH.md "rocprofv3 PMC slots"), --kernel-trace only.  usage: tools/gpu_traffic.sh <outdir> [corpus] [mb]
set -u
out="$1"; corpus="${2:-english}"; mb="${3:-256}"
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
rm -rf "$R/gpurun_out/$out"; mkdir -p "$R/gpurun_out/$out"   # never mix the CSVs of two runs
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$R/gpurun_out/$out/$c" -- python "$R/tools/gpu_ablate.py" $corpus $mb 0 > "$R/gpurun_out/$out/$c.log" 2>&1
done
find "$R/gpurun_out/$out" -name "*counter_collection.csv" | head
