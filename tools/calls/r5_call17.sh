cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_small; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_python_api.py tests/test_gpu_char_seeds.py tests/test_property_based.py -x -q -m gpu > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 300 python tools/gpu_latency.py > $O/latency.txt 2>&1; grep -v amdgpu $O/latency.txt | grep "repetitive\|whitespace\|emoji\|mixed corpus\|hello\|minimal: empty\|DIFFER" | cut -c1-110
