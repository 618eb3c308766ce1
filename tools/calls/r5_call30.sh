cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_e2e; mkdir -p $O
( for b in 8 16 32 64; do echo "== TD_PIPE_D2H_KERNEL=$b"; TD_PIPE_D2H_KERNEL=$b timeout 300 python tools/gpu_e2e_sweep.py 2>&1 | grep -v amdgpu.ids | tail -3; done
echo "== TD_PIPE_D2H_KERNEL=32 TD_PIPE_STREAMS=2"; TD_PIPE_STREAMS=2 TD_PIPE_D2H_KERNEL=32 timeout 300 python tools/gpu_e2e_sweep.py 2>&1 | grep -v amdgpu.ids | tail -3 ) > $O/sweep5.txt; cat $O/sweep5.txt
