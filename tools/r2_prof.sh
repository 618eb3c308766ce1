set -u
# usage: tools/r2_prof.sh <tag> [corpora...]: rocprofv3 kernel stats of tools/gpu_ablate.py per corpus -> gpurun_out/<tag>/stats_<corpus>.txt
R=$GRAFT_REPO_ROOT; tag=$1; shift; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for c in "$@"; do
  rm -rf $O/stats_$c
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats_$c -- python $R/tools/gpu_ablate.py $c 256 0 > $O/ab_$c.log 2>&1
  grep stop_after $O/ab_$c.log
  python $R/tools/prof_summary.py $O/stats_$c $O/stats_${c}.txt "$c 256 MiB (tools/gpu_ablate.py)" > /dev/null
  head -16 $O/stats_${c}.txt
  rm -rf $O/stats_$c
done
