cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_final10; mkdir -p $O
timeout 80 python bench.py --no-cpu-baseline --no-side-configs --steps 50 > $O/bench_final_sources_short.json 2> $O/bench.err; python - $O/bench_final_sources_short.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = j["roofline"]
print(j["value"], j["ms_per_step"], j["config"]["verified_vs_oracle"], "traffic", r.get("traffic"), "counters", bool(r.get("counters")))
print(str(r.get("traffic_note"))[-260:])
PY
