cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_b1; mkdir -p $O
( time timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time; tail -3 $O/bench_default.time
tail -5 $O/bench_default.err
python - "$O/bench_default.json" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = j["roofline"]
print(f'headline {j["value"]:.1f} GB/s  {j["ms_per_step"]:.4f} ms  plain {j["launch_mode"]}  dominant {r["kernel"]} {r["kernel_ms_avg"]} ms frac {r["frac"]}  whole-step frac {r["whole_step"]["frac"]}  fixed {r.get("fixed_overhead_us")} us  verified {j["config"]["verified_vs_oracle"]}')
print('cpu_baseline', j.get("cpu_baseline", {}).get("value"), 'python', (j.get("drop_in_python") or {}).get("threads_8"), 'x', (j.get("drop_in_python") or {}).get("vs_reference_encode_batch"))
print('e2e', j.get("e2e"))
for k, v in (j.get("configs") or {}).items():
    print(f'  {k:<34} {v.get("value")} GB/s (no dedupe {v.get("value_without_dedupe")})  {v.get("ms_per_step")} ms  {v.get("verified_vs_oracle", v.get("error"))}  ({v.get("wall_s")} s)')
PY
