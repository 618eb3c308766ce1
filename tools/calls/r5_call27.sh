cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_e2e; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pipelined or host" 2>&1 | tail -2
TD_PIPE_TIMING=1 timeout 400 python tools/gpu_e2e_sweep.py 2>&1 | grep -v amdgpu.ids > $O/sweep2.txt; grep -v pipeline: $O/sweep2.txt; grep pipeline: $O/sweep2.txt | awk 'NR%4==0' | head -20
