cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_final9; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_python_api.py -x -q -m gpu 2>&1 | tail -2
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
TD_PIPE_TIMING=1 timeout 300 python tools/gpu_e2e_sweep.py 2>&1 | grep -v amdgpu.ids | tail -2
