set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( bash tools/r2_ab.sh "cur" "mixed code_files" 12,13,0
  bash tools/r2_ab.sh "cur" "english" 0 ) > gpurun_out/ab14.txt 2>&1
cat gpurun_out/ab14.txt
