cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_ovt; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_overlap.py -x -q -m gpu > $O/pytest.log 2>&1; tail -6 $O/pytest.log
