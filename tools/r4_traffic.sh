#!/bin/bash
# HBM traffic of one bench workload (two --pmc passes; bytes per launch do not depend on the step count):
# tools/r4_traffic.sh <out name> [bench args...]    -> gpurun_out/<out name>/{FETCH_SIZE,WRITE_SIZE}/..., summary on stdout
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; name=$1; shift
O=$R/gpurun_out/$name; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-verify --steps 3 --warmup 1 $*"
for c in FETCH_SIZE WRITE_SIZE; do rm -rf $O/$c; TD_BENCH_GRAPH=0 timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$c -- $B > /dev/null 2> $O/$c.err; done
find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete
python $R/tools/pmc_summary.py $O raw
