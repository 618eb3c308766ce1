set -u
# usage: tools/prof_workload.sh <tag> <corpus> <MiB> [bench args]: rocprofv3 kernel stats of the bench command -> gpurun_out/<tag>/stats_<corpus>_<MiB>.txt (+ the raw .db)
R=$GRAFT_REPO_ROOT; tag=$1; c=$2; mb=$3; shift 3; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --corpus $c --size-mb $mb --no-cpu-baseline --no-verify --steps 10 --warmup 3 $*"
rm -rf $O/stats_${c}_$mb
timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats_${c}_$mb -- $B > $O/stats_${c}_$mb.json 2> $O/stats_${c}_$mb.err
python $R/tools/prof_summary.py $O/stats_${c}_$mb $O/stats_${c}_$mb.txt "rocprofv3 --kernel-trace --stats -- $B" > /dev/null
head -20 $O/stats_${c}_$mb.txt
find $O/stats_${c}_$mb -name "*.csv" -delete
