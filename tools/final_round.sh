set -u
# Closing run of a round on the final sources (ONE gpurun call): the full GPU suite FIRST (what the driver runs, same flags),
# then the default bench line (what the driver records), then the profiles the numbers in DESIGN.md come from.
# usage: tools/final_round.sh <tag>      -> gpurun_out/<tag>/ ; publish with python tools/publish_round.py <tag> <round>
R=$GRAFT_REPO_ROOT; tag=$1; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
nproc > $O/host.txt; grep -m1 "model name" /proc/cpuinfo >> $O/host.txt
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time; tail -3 $O/bench_default.time
python - "$O/bench_default.json" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = j["roofline"]
print(f'headline {j["value"]:.1f} GB/s  {j["ms_per_step"]:.4f} ms  dominant {r["kernel"]} {r["kernel_ms_avg"]} ms frac {r["frac"]}  whole-step frac {r["whole_step"]["frac"]}  fixed {r.get("fixed_overhead_us")} us  verified {j["config"]["verified_vs_oracle"]}')
print('cpu_baseline', j.get("cpu_baseline", {}).get("value"), 'python', (j.get("drop_in_python") or {}).get("threads_8"), 'x', (j.get("drop_in_python") or {}).get("vs_reference_encode_batch"))
for k, v in (j.get("configs") or {}).items():
    print(f'  {k:<34} {v.get("value")} GB/s  {v.get("ms_per_step")} ms  {v.get("verified_vs_oracle", v.get("error"))}  ({v.get("wall_s")} s)')
PY
timeout 400 python bench.py --gpus 2 --same-gpu --dist-backend gloo --no-cpu-baseline --steps 30 > $O/bench_2rank_same_gpu_gloo.json 2> $O/bench_2rank.err; python -c "
import json,sys; j=json.loads(open('$O/bench_2rank_same_gpu_gloo.json').read().strip().splitlines()[-1]); print('2 ranks on one GPU (gloo):', j['value'], 'GB/s', j['config']['verified_vs_oracle'], j['config']['parallelism'][:60])" 2>&1 | tail -1
bash tools/measure_workload.sh $tag english 1024 > $O/measure_english.log 2>&1; tail -14 $O/measure_english.log
# (fabric traffic of configs 4 and 5 too, on the shipping sources; these also leave stats_<corpus>_256.txt)
bash tools/measure_workload.sh $tag mixed 256 --pattern tekken > $O/measure_mixed_tekken.log 2>&1; tail -4 $O/measure_mixed_tekken.log; mv $O/stats_mixed_256.txt $O/stats_mixed_tekken_256.txt
bash tools/measure_workload.sh $tag mixed 256 > $O/measure_mixed.log 2>&1; tail -4 $O/measure_mixed.log
bash tools/measure_workload.sh $tag code_files 256 > $O/measure_code_files.log 2>&1; tail -4 $O/measure_code_files.log
for cm in "english 1024" "mixed 256" "code_files 256"; do set -- $cm; bash tools/pmc_workload.sh $tag $1 $2 > /dev/null 2>&1; head -12 $O/pmc_$1_$2.txt | cut -c1-230; rm -rf $O/pmc_$1_$2; done
cp $R/profiles/hbm_traffic.json $O/hbm_traffic.json
timeout 200 python tools/gpu_pybatch.py 256 > $O/pybatch.txt 2>&1; grep -v amdgpu $O/pybatch.txt
timeout 300 python tools/gpu_latency.py > $O/latency.txt 2>&1; grep -v amdgpu $O/latency.txt | tail -12
timeout 600 python tools/gpu_latency_files.py > $O/latency_files.txt 2>&1; grep -v amdgpu $O/latency_files.txt | tail -25
timeout 200 python tools/gpu_giant.py > $O/giant_pieces.txt 2>&1; grep -v amdgpu $O/giant_pieces.txt | tail -16
# (the direct placement against the staged form on the final sources, same box and corpus)
timeout 300 python tools/gpu_opt_ab.py english 1024 20 DIRECT=1,0 > $O/direct_ab.txt 2>&1; grep -v amdgpu $O/direct_ab.txt | tail -12
# (the sparse launch sequence against the dense one on plain text, 128 MiB and 1024 MiB)
for mb in 128 1024; do for sp in 1 0; do TD_SPARSE=$sp timeout 300 python bench.py --corpus english --size-mb $mb --no-cpu-baseline --no-side-configs --steps 50 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('english $mb MiB, TD_SPARSE=$sp:', j['value'], 'GB/s', j['ms_per_step'], 'ms', j['launch_mode']['launch_sequence'], 'plain launches', j['launch_mode']['ms_per_step_plain_launches'], 'ms; graph replay', j['launch_mode']['ms_per_step_graph_replay'], 'ms')"; done; done > $O/sequences_ab.txt 2>&1; cat $O/sequences_ab.txt
find $O -name "*.db" -size +20M -delete
