set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_python_api.py tests/test_tekken_pattern.py -m gpu -x -q 2>&1 | tail -5
bash tools/r2_ab.sh "prev cur" "mixed code_files english" 
bash tools/r2_ab.sh "cur" "mixed english" 12 ) > gpurun_out/ab16.txt 2>&1
cat gpurun_out/ab16.txt
