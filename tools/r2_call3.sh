set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2_c3; mkdir -p $O
cd $R
for c in english mixed; do
  timeout 300 python tools/gpu_ablate.py $c 256 12,2,30,31,3,0 2>&1 | grep stop_after
done | tee $O/ablate.txt
