cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_giant; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_digit_runs.py -x -q -m gpu -s 2>&1 | grep -v "^$" | tail -40 > $O/pytest_digits.txt
tail -30 $O/pytest_digits.txt
