set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash tools/r2_ab.sh "base fast fastonly" "english mixed code" > gpurun_out/ab12.txt 2>&1
for v in fast fastonly; do
  TD_HIP_LIB=$GRAFT_REPO_ROOT/variants/$v.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 | sed "s/^/$v /" >> gpurun_out/ab12.txt
done
cat gpurun_out/ab12.txt
