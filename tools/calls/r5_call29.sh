cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_e2e; mkdir -p $O
( echo "== HSA_ENABLE_SDMA_GANG=0"; HSA_ENABLE_SDMA_GANG=0 TD_PIPE_TIMING=1 timeout 300 python tools/gpu_e2e_sweep.py 2>&1 | grep -v amdgpu.ids | tail -2
for b in 64 256 1024; do echo "== TD_PIPE_D2H_KERNEL=$b"; TD_PIPE_D2H_KERNEL=$b TD_PIPE_TIMING=1 timeout 300 python tools/gpu_e2e_sweep.py 2>&1 | grep -v amdgpu.ids | tail -2; done ) > $O/sweep4.txt; cat $O/sweep4.txt
