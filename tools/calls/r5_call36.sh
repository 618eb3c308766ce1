cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_final10; mkdir -p $O
cat /sys/kernel/mm/transparent_hugepage/enabled
timeout 50 python tools/gpu_pybatch.py 256 2>&1 | grep -v amdgpu > $O/pybatch.txt; cat $O/pybatch.txt
