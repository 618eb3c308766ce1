set -u
R=$GRAFT_REPO_ROOT; cd $R
for v in norounds; do
  for c in mixed code; do
  TD_HIP_LIB=$R/variants/$v.so timeout 300 python tools/gpu_ablate.py $c 256 0 2>&1 | grep stop_after | sed "s/^/$v /"
  done
done
