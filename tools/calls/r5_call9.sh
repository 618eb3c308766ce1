cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_ml; mkdir -p $O
for ml in 2 9; do
for c in "mixed none" "code_files none" "chat all"; do set -- $c
TD_DD_MINLEN=$ml timeout 300 python bench.py --corpus $1 --allowed-special $2 --size-mb 256 --no-cpu-baseline --no-verify --steps 20 --warmup 3 > $O/b.json 2> $O/b.err
python - $O/b.json $ml $1 <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = j["roofline"]
print("minlen", sys.argv[2], sys.argv[3], j["value"], "GB/s", j["ms_per_step"], "ms", {k.split("+")[0].replace("td_", ""): v for k, v in r["all_kernels_ms_avg"].items()})
PY
done; done
