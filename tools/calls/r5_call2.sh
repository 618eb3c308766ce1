cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_dd; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_dedupe.py tests/test_gpu_fused.py -x -q -m gpu > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 300 python tools/gpu_opt_ab.py mixed,code_files,english 256 10 DEDUPE=1,0 > $O/ab.txt 2>&1; grep -v amdgpu $O/ab.txt | tail -20
