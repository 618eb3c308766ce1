set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2_c5; mkdir -p $O
cd $R
for v in nb1 nb2 nb2w6 nb4w6; do
  for c in english code; do
  TD_HIP_LIB=$R/variants/$v.so timeout 300 python tools/gpu_ablate.py $c 256 3 2>&1 | grep stop_after | sed "s/^/$v /"
  done
done | tee $O/ablate.txt
