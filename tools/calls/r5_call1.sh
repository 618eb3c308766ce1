cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5_pmc
rocprofv3 -L > gpurun_out/r5_pmc/counters_available.txt 2>&1 || true
bash tools/pmc_workload.sh r5_pmc english 1024 > /dev/null 2>&1
bash tools/pmc_workload.sh r5_pmc mixed 256 > /dev/null 2>&1
bash tools/pmc_workload.sh r5_pmc code_files 256 > /dev/null 2>&1
head -30 gpurun_out/r5_pmc/pmc_english_1024.txt
