cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_cw; mkdir -p $O
for lib in "" variants/cw8.so variants/cw4.so; do
for c in "mixed none" "code_files none"; do set -- $c
TD_HIP_LIB=$lib TD_OVERLAP=0 timeout 300 python bench.py --corpus $1 --allowed-special $2 --size-mb 256 --no-cpu-baseline --no-verify --steps 20 --warmup 3 > $O/b.json 2> $O/b.err
python - $O/b.json $1 "$lib" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = j["roofline"]
print("lib", sys.argv[3] or "default(6)", sys.argv[2], j["value"], "GB/s", j["ms_per_step"], "ms", {k.split("+")[0].replace("td_", ""): v for k, v in r["all_kernels_ms_avg"].items() if "merge" in k})
PY
done; done
