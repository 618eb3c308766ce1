cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_e2e; mkdir -p $O
for v in "1 1" "0 1" "1 2" "0 2"; do set -- $v; echo "== TD_PIPE_PUBLISH=$1 TD_PIPE_STREAMS=$2"; TD_PIPE_PUBLISH=$1 TD_PIPE_STREAMS=$2 TD_PIPE_TIMING=1 timeout 300 python tools/gpu_e2e_sweep.py 2>&1 | grep -v amdgpu.ids | tail -2; done > $O/sweep3.txt; cat $O/sweep3.txt
