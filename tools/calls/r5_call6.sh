cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_cs; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py tests/test_gpu_dedupe.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for cs in 1 0; do
TD_CHAR_SEEDS=$cs timeout 300 python bench.py --corpus mixed --size-mb 256 --no-cpu-baseline --steps 20 --warmup 3 > $O/bench_mixed_cs$cs.json 2> $O/bench_mixed_cs$cs.err
python - $O/bench_mixed_cs$cs.json $cs <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = j["roofline"]
print("char seeds", sys.argv[2], j["value"], "GB/s", j["ms_per_step"], "ms", j["config"]["verified_vs_oracle"], {k: v for k, v in r["all_kernels_ms_avg"].items()})
PY
done
bash tools/prof_workload.sh r5_cs mixed 256 > $O/prof_mixed.log 2>&1; head -12 $O/stats_mixed_256.txt | cut -c1-120
find $O -name "*.db" -delete
