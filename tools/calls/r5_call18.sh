cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_c2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dedupe.py tests/test_gpu_fused.py tests/test_gpu_parity.py -x -q -m gpu > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for c in "mixed none" "code_files none" "chat all" "english none"; do set -- $c
TD_OVERLAP=0 timeout 300 python bench.py --corpus $1 --allowed-special $2 --size-mb 256 --no-cpu-baseline --no-verify --steps 20 --warmup 3 > $O/b.json 2> $O/b.err
python - $O/b.json $1 <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = j["roofline"]
print(sys.argv[2], j["value"], "GB/s", j["ms_per_step"], "ms", {k.split("+")[0].replace("td_", ""): v for k, v in r["all_kernels_ms_avg"].items()})
PY
done
