cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_st; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dedupe.py tests/test_gpu_char_seeds.py tests/test_abi.py -x -q -m gpu > $O/pytest.log 2>&1; tail -6 $O/pytest.log
