cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_giant; mkdir -p $O
for g in 256 128 64; do echo "== TD_GIANT_BLOCKS=$g"; TD_GIANT_BLOCKS=$g timeout 300 python tools/gpu_giant.py 2>&1 | grep -v amdgpu.ids; done > $O/giant_table.txt
cat $O/giant_table.txt
