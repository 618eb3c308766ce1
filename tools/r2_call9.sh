set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2_c9; mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
bash tools/r2_prof.sh r2_c9 english mixed code 2>&1 | grep -v "^#\|^kernel\|amd_rocclr\|td_prepare\|td_mark"
