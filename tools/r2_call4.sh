set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2_c4; mkdir -p $O
cd $R
for v in diag_sameslot probe8 probe5; do
  TD_HIP_LIB=$R/variants/$v.so timeout 300 python tools/gpu_ablate.py english 256 31,3 2>&1 | grep stop_after | sed "s/^/$v /"
done | tee $O/ablate.txt
