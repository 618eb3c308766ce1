set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2_c11; mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1
tail -4 $O/pytest.log
( time timeout 900 python bench.py ) > $O/bench_english_1024.json 2> $O/bench_english_1024.err
cat $O/bench_english_1024.json; tail -2 $O/bench_english_1024.err
for c in mixed code code_files; do
 timeout 900 python bench.py --corpus $c --size-mb 256 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_${c}_256.json 2> $O/bench_${c}_256.err; python -c "
import json;d=json.load(open('$O/bench_${c}_256.json'));print('$c',d['value'],'GB/s',d['ms_per_step'],'ms',d['config']['verified_vs_oracle'],d['roofline']['all_kernels_ms_avg'])"; tail -1 $O/bench_${c}_256.err
done
timeout 900 python bench.py --corpus mixed --pattern tekken --size-mb 256 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_mixed_tekken_256.json 2> $O/bench_mixed_tekken_256.err; python -c "
import json;d=json.load(open('$O/bench_mixed_tekken_256.json'));print('tekken',d['value'],'GB/s',d['config']['verified_vs_oracle'])"
timeout 600 python bench.py --gpus 2 --same-gpu --dist-backend gloo --size-mb 256 --no-cpu-baseline --steps 5 --warmup 2 > $O/bench_2rank.json 2> $O/bench_2rank.err; python -c "
import json;d=json.load(open('$O/bench_2rank.json'));print('2rank',d['value'],'GB/s n_gpus',d['n_gpus'],d['config']['verified_vs_oracle'])"
