set -u
# The parts of tools/measure_round2.sh that depend on the encode kernels only (bench lines, kernel stats, HBM traffic):
# for a last refresh after a kernel change when the SQ counter passes, the host-path rates and the multi-rank lines stand.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2_final; mkdir -p $O
cd $R
timeout 900 python bench.py > $O/bench_english_1024.json 2> $O/bench_english_1024.err
timeout 600 python bench.py --size-mb 256 --no-cpu-baseline > $O/bench_english_256.json 2> $O/bench_english_256.err
for c in mixed code code_files; do timeout 600 python bench.py --corpus $c --size-mb 256 --steps 5 --warmup 2 > $O/bench_${c}_256.json 2> $O/bench_${c}_256.err; done
timeout 600 python bench.py --corpus mixed --pattern tekken --size-mb 256 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_mixed_tekken_256.json 2> $O/bench_mixed_tekken_256.err
timeout 900 python bench.py --corpus mixed --size-mb 1024 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_mixed_1024.json 2> $O/bench_mixed_1024.err
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-verify --steps 10 --warmup 3"
rm -rf $O/stats; timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats -- $B > $O/stats_bench.json 2> $O/stats.err
python $R/tools/prof_summary.py $O/stats $O/stats_english_1024.txt "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-verify --steps 10 --warmup 3 (1024 MiB English)" > /dev/null
for c in mixed code code_files; do timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats_$c -- python $R/bench.py --corpus $c --size-mb 256 --no-cpu-baseline --no-verify --steps 5 --warmup 2 > /dev/null 2> $O/stats_$c.err; python $R/tools/prof_summary.py $O/stats_$c $O/stats_${c}_256.txt "rocprofv3 --kernel-trace --stats -- python bench.py --corpus $c --size-mb 256 --no-cpu-baseline --no-verify --steps 5 --warmup 2" > /dev/null; rm -rf $O/stats_$c; done
rm -rf $O/stats
for c in FETCH_SIZE WRITE_SIZE; do rm -rf $O/traffic_english/$c; timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/traffic_english/$c -- $B > /dev/null 2> $O/traffic_$c.err; done
for w in mixed code; do for c in FETCH_SIZE WRITE_SIZE; do rm -rf $O/traffic_$w/$c; timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/traffic_$w/$c -- python $R/bench.py --corpus $w --size-mb 256 --no-cpu-baseline --no-verify --steps 5 --warmup 2 > /dev/null 2>> $O/traffic_$w.err; done; done
find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete
ls $O | head -3
