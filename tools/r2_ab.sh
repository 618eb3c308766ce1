set -u
# usage: tools/r2_ab.sh "<variants>" "<corpora>" [stops]
R=$GRAFT_REPO_ROOT; cd $R
for v in $1; do
  for c in $2; do
    lib=$R/variants/$v.so; [ "$v" = cur ] && lib=$R/tokendagger_amd/libtokendagger_hip.so
    TD_HIP_LIB=$lib timeout 300 python tools/gpu_ablate.py $c 256 ${3:-0} 2>&1 | grep stop_after | sed "s/^/$v /"
  done
done
