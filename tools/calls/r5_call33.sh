cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_giant; mkdir -p $O
timeout 500 python -m pytest tests/test_gpu_giant_coop.py "tests/test_gpu_parity.py::test_giant_pieces_are_not_quadratic" -x -q -m gpu -s 2>&1 | grep -v "^$" | tail -14
