set -u
# HBM traffic of the headline workload only (two --pmc passes, few steps: bytes per launch do not depend on the step count)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2_final; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-verify --steps 3 --warmup 1"
for c in FETCH_SIZE WRITE_SIZE; do rm -rf $O/traffic_english/$c; timeout 100 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/traffic_english/$c -- $B > /dev/null 2> $O/traffic_$c.err; done
find $O/traffic_english -name "*.db" -delete; find $O/traffic_english -name "*agent_info.csv" -delete
ls $O/traffic_english/*/*/ | head -4
