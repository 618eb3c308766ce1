cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_gpu_overlap.py tests/test_gpu_clone.py tests/test_gpu_giant_coop.py tests/test_gpu_digit_runs.py tests/test_gpu_dedupe.py tests/test_gpu_fused.py -x -q -m gpu 2>&1 | tail -2
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
