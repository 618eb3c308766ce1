O=$GRAFT_REPO_ROOT/gpurun_out/r5_cs3; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for cs in 1 0; do for c in code_files mixed; do
TD_CHAR_SEEDS=$cs TD_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O/st_${c}_$cs -- python $GRAFT_REPO_ROOT/bench.py --corpus $c --size-mb 256 --no-cpu-baseline --no-verify --steps 10 --warmup 3 > $O/b_${c}_$cs.json 2> $O/st.err
python $GRAFT_REPO_ROOT/tools/prof_summary.py $O/st_${c}_$cs $O/st_${c}_$cs.txt "$c seeds $cs" > /dev/null; echo "== $c seeds=$cs"; grep "long_pieces\|merge_pieces\|collect\|copy_dups" $O/st_${c}_$cs.txt | cut -c1-110
python -c "
import json; j=json.loads(open('$O/b_${c}_$cs.json').read().strip().splitlines()[-1]); print(j['value'], 'GB/s', j['ms_per_step'])"
done; done
find $O -name "*.db" -delete; find $O -name "*.csv" -delete
