set -u
# Measurement run (one gpurun call): bench line of the headline config, rocprofv3 kernel stats of the bench command,
# HBM traffic (separate --pmc passes, --kernel-trace only).  usage: tools/measure_workload.sh <tag> [corpus] [MiB] [extra bench args]
R=$GRAFT_REPO_ROOT; tag=$1; c=${2:-english}; mb=${3:-1024}; shift; shift || true; shift || true
O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
B="python $R/bench.py --corpus $c --size-mb $mb --no-cpu-baseline --no-verify --steps 10 --warmup 3 $*"
cd /tmp; export TMPDIR=/tmp
rm -rf $O/stats_${c}_$mb
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_${c}_$mb -- $B > $O/stats_${c}_$mb.json 2> $O/stats_${c}_$mb.err
python $R/tools/prof_summary.py $O/stats_${c}_$mb $O/stats_${c}_$mb.txt "rocprofv3 --kernel-trace --stats -- $B" > /dev/null
find $O/stats_${c}_$mb -name "*.csv" -delete
for k in FETCH_SIZE WRITE_SIZE; do rm -rf $O/traffic_${c}_$mb/$k; timeout 300 rocprofv3 --kernel-trace --pmc $k --output-format csv -d $O/traffic_${c}_$mb/$k -- $B > /dev/null 2> $O/traffic_${c}_${mb}_$k.err; done
pat=llama4; case " $* " in *" --pattern tekken "*) pat=tekken;; esac
python $R/tools/update_traffic.py ${c}_${pat}_$mb $O/traffic_${c}_$mb "gpurun_out/$tag, tools/measure_workload.sh" $O/stats_${c}_$mb.json > $O/traffic_${c}_$mb.txt 2>&1
cp $R/profiles/hbm_traffic.json $O/hbm_traffic.json
find $O -name "*agent_info.csv" -delete
cat $O/stats_${c}_$mb.txt | head -16; cat $O/traffic_${c}_$mb.txt
