cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r5_chat; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for ml in 2 9; do
TD_DD_MINLEN=$ml timeout 300 rocprofv3 --kernel-trace --stats -d $O/st$ml -- python $GRAFT_REPO_ROOT/bench.py --corpus chat --allowed-special all --size-mb 256 --no-cpu-baseline --no-verify --steps 10 --warmup 3 > /dev/null 2> $O/st$ml.err
python $GRAFT_REPO_ROOT/tools/prof_summary.py $O/st$ml $O/st$ml.txt "chat minlen $ml" > /dev/null; head -14 $O/st$ml.txt | cut -c1-110
done
find $O -name "*.db" -delete; find $O -name "*.csv" -delete
