set -u
# Round-3 closing run on the final sources (one gpurun call): the full GPU suite, then the measurement passes.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_final; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q --timeout 400 > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
bash tools/r3_measure.sh r3_final english 1024 > $O/measure_english.log 2>&1; tail -12 $O/measure_english.log
bash tools/r3_measure.sh r3_final mixed 256 > $O/measure_mixed.log 2>&1; tail -4 $O/measure_mixed.log
bash tools/r3_measure.sh r3_final code_files 256 > $O/measure_code_files.log 2>&1; tail -4 $O/measure_code_files.log
bash tools/r3_prof.sh r3_final chat 256 --allowed-special all > $O/prof_chat.log 2>&1; tail -4 $O/prof_chat.log
cp $R/profiles/hbm_traffic.json $O/hbm_traffic.json
