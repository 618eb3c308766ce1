cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_final9; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time; tail -3 $O/bench_default.time
python - "$O/bench_default.json" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = j["roofline"]
print(f'headline {j["value"]:.1f} GB/s  {j["ms_per_step"]:.4f} ms  traffic {r.get("traffic") is not None} counters {r.get("counters") is not None}')
print('e2e', j.get("e2e", {}).get("value"), j.get("e2e", {}).get("frac_of_pcie_ceiling"), 'python', (j.get("drop_in_python") or {}).get("vs_reference_encode_batch"), (j.get("drop_in_python") or {}).get("vs_reference_encode_batch_2560_chunks"))
for k, v in (j.get("configs") or {}).items():
    print(f'  {k:<34} {v.get("value")} GB/s  {v.get("verified_vs_oracle", v.get("error"))}')
PY
timeout 200 python tools/gpu_pybatch.py 256 > $O/pybatch.txt 2>&1; grep -v amdgpu $O/pybatch.txt
TD_PIPE_TIMING=1 timeout 300 python tools/gpu_e2e_sweep.py 2>&1 | grep -v amdgpu.ids > $O/e2e_sweep_final.txt; grep -v pipeline: $O/e2e_sweep_final.txt
