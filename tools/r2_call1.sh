set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2_c1; mkdir -p $O
cd $R
nproc > $O/host.txt; grep -m1 "model name" /proc/cpuinfo >> $O/host.txt
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
( time timeout 900 python bench.py ) > $O/bench_english_1024.json 2> $O/bench_english_1024.err
cat $O/bench_english_1024.json; tail -3 $O/bench_english_1024.err
( time timeout 600 python bench.py --gpus 2 --same-gpu --dist-backend gloo --size-mb 256 --no-cpu-baseline --steps 5 --warmup 2 ) > $O/bench_2rank_same_gpu.json 2> $O/bench_2rank_same_gpu.err
cat $O/bench_2rank_same_gpu.json; tail -5 $O/bench_2rank_same_gpu.err
TD_BENCH_FORCE_DIST=1 timeout 600 python bench.py --size-mb 256 --no-cpu-baseline --steps 5 --warmup 2 > $O/bench_dist1.json 2> $O/bench_dist1.err
cat $O/bench_dist1.json; tail -3 $O/bench_dist1.err
