set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2_c7; mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_python_api.py -m gpu -x -q ) > $O/pytest_a.log 2>&1
tail -8 $O/pytest_a.log
bash tools/r2_prof.sh r2_c7 english mixed code 2>&1 | grep -v "^#\|^kernel\|amd_rocclr\|td_prepare\|td_mark\|split_slow"
