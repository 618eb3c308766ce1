set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2_c2; mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_python_api.py -m gpu -x -q ) > $O/pytest_a.log 2>&1
tail -15 $O/pytest_a.log
for c in english mixed code; do
  TD_HIP_LIB=$R/variants/r2_base.so timeout 300 python tools/gpu_ablate.py $c 256 0 2>&1 | tail -2 | sed "s/^/base /"
  timeout 300 python tools/gpu_ablate.py $c 256 0 2>&1 | tail -2 | sed "s/^/new  /"
done | tee $O/ab.txt
