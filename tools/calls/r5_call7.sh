cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_cs2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_char_seeds.py tests/test_gpu_parity.py tests/test_gpu_fused.py tests/test_gpu_dedupe.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/pytest.log 2>&1; tail -5 $O/pytest.log
