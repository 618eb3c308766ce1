set -u
# Round-3 bench lines of every BASELINE config (one gpurun call).  usage: tools/r3_lines.sh <tag> [regex of line names to run]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
cd $R
nproc > $O/host.txt; grep -m1 "model name" /proc/cpuinfo >> $O/host.txt
ONLY=${2:-.}
run() { name=$1; shift; echo "$name" | grep -Eq "$ONLY" || return 0; timeout 400 python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err || echo "FAILED $name"; python - "$O/bench_$name.json" "$name" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=j["roofline"]
    print(f'{sys.argv[2]:<24} {j["value"]:8.1f} GB/s  {j["ms_per_step"]:8.3f} ms  whole-step frac {r["whole_step"]["frac"]:.4f}  verified {j["config"]["verified_vs_oracle"]}  fixed {r.get("fixed_overhead_us")}')
except Exception as e:
    print(sys.argv[2], "no line:", e)
PY
}
run english_1024
run english_256 --size-mb 256 --no-cpu-baseline
run mixed_256 --corpus mixed --size-mb 256 --steps 30 --no-cpu-baseline
run mixed_1024 --corpus mixed --size-mb 1024 --steps 20 --no-cpu-baseline
run mixed_tekken_256 --corpus mixed --pattern tekken --size-mb 256 --steps 30 --no-cpu-baseline
run mixed_tekken_1024 --corpus mixed --pattern tekken --size-mb 1024 --steps 20
run code_256 --corpus code --size-mb 256 --steps 30 --no-cpu-baseline
run code_files_256 --corpus code_files --size-mb 256 --steps 30 --no-cpu-baseline
run code_files_1024 --corpus code_files --size-mb 1024 --steps 20 --no-cpu-baseline
run chat_specials_256 --corpus chat --allowed-special all --size-mb 256 --steps 30 --no-cpu-baseline
run chat_specials_1024 --corpus chat --allowed-special all --size-mb 1024 --steps 20 --no-cpu-baseline
run generic_autogen_256 --pattern generic:autogen --size-mb 256 --steps 10 --warmup 2 --no-cpu-baseline
run generic_autogen_64_single_document --pattern generic:autogen --size-mb 64 --single-document --steps 10 --warmup 2 --no-cpu-baseline
run generic_autogen_64_many_documents --pattern generic:autogen --size-mb 64 --steps 10 --warmup 2 --no-cpu-baseline
run 2rank_same_gpu_gloo_1024 --gpus 2 --same-gpu --dist-backend gloo --no-cpu-baseline --steps 30
TD_BENCH_FORCE_DIST=1 run rccl_world1_torch_1024 --no-cpu-baseline --steps 50
TD_BENCH_FORCE_DIST=1 run rccl_world1_capi_1024 --no-cpu-baseline --steps 50 --collective capi
[ "$ONLY" = . ] || exit 0
timeout 300 python tools/gpu_hostpath.py > $O/hostpath.txt 2>&1; tail -12 $O/hostpath.txt
timeout 100 python tools/gpu_latency.py > $O/latency.txt 2>&1; cat $O/latency.txt | tail -3
