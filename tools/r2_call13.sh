set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_python_api.py tests/test_cl100k_pattern.py tests/test_gpt2_pattern.py tests/test_tekken_pattern.py -m gpu -x -q 2>&1 | tail -5
bash tools/r2_ab.sh "fastonly cur" "english mixed code code_files" 
bash tools/r2_ab.sh "cur" "english" 12,13 ) > gpurun_out/ab13.txt 2>&1
cat gpurun_out/ab13.txt
