cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_ov; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py tests/test_gpu_char_seeds.py tests/test_gpu_clone.py tests/test_gpu_special_device.py -x -q -m gpu > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 300 python tools/gpu_opt_ab.py mixed,code_files,english 256 10 OVERLAP=1,0 > $O/ab.txt 2>&1; grep -v amdgpu $O/ab.txt | cut -c1-60,170-230 | tail -15
