cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_e2e; mkdir -p $O
timeout 400 python tools/gpu_e2e_sweep.py 2>&1 | grep -v amdgpu.ids > $O/sweep.txt; cat $O/sweep.txt
