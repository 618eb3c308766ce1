cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_dd2; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_dedupe.py tests/test_gpu_fused.py -x -q -m gpu > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 300 python tools/gpu_opt_ab.py mixed,code_files,english 256 10 DEDUPE=1,0 > $O/ab.txt 2>&1; grep -v amdgpu $O/ab.txt | grep -v "^english.*DEDUPE" | cut -c1-250 | tail -12
bash tools/prof_workload.sh r5_dd2 mixed 256 > $O/prof_mixed.log 2>&1; head -14 $O/stats_mixed_256.txt
bash tools/prof_workload.sh r5_dd2 code_files 256 > $O/prof_cf.log 2>&1; head -14 $O/stats_code_files_256.txt
find $O -name "*.db" -delete
