cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_e2e; mkdir -p $O
( timeout 200 python tools/gpu_pcie_duplex.py; HSA_ENABLE_SDMA=0 timeout 200 python tools/gpu_pcie_duplex.py ) 2>&1 | grep -v amdgpu.ids > $O/duplex.txt; cat $O/duplex.txt
