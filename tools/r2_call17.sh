set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_python_api.py tests/test_tekken_pattern.py -m gpu -x -q 2>&1 | tail -3
bash tools/r2_ab.sh "prev cur" "code_files mixed english" ) > gpurun_out/ab17.txt 2>&1
cat gpurun_out/ab17.txt
