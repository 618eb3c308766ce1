set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r1_final; mkdir -p $O
cd $R
nproc > $O/host.txt; grep -m1 "model name" /proc/cpuinfo >> $O/host.txt
timeout 600 python bench.py > $O/bench_english_256.json 2> $O/bench_english_256.err
timeout 600 python bench.py --size-mb 1024 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_english_1024.json 2> $O/bench_english_1024.err
timeout 600 python bench.py --corpus code --steps 5 --warmup 2 > $O/bench_code_256.json 2> $O/bench_code_256.err
timeout 600 python bench.py --corpus mixed --steps 5 --warmup 2 > $O/bench_mixed_256.json 2> $O/bench_mixed_256.err
timeout 600 python bench.py --corpus mixed --pattern tekken --steps 5 --warmup 2 > $O/bench_mixed_tekken_256.json 2> $O/bench_mixed_tekken_256.err
TD_BENCH_FORCE_DIST=1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 python bench.py --no-cpu-baseline > $O/bench_dist1.json 2> $O/bench_dist1.err
timeout 300 python tools/gpu_e2e.py > $O/e2e.txt 2>&1
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats -- python $R/bench.py --no-cpu-baseline --no-verify > $O/stats_bench.json 2> $O/stats.err
python $R/tools/prof_summary.py $O/stats $O/stats_summary.txt "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-verify (256 MiB English, 10 steps + 3 warmup)" > /dev/null
for c in code mixed; do timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_$c -- python $R/tools/gpu_ablate.py $c 256 0 > $O/ab_$c.log 2>&1; python $R/tools/prof_summary.py $O/stats_$c $O/stats_${c}_summary.txt "$c 256 MiB (tools/gpu_ablate.py)" > /dev/null; done
