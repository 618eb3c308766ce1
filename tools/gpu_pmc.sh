#!/usr/bin/env bash
# PMC passes for the fused tile kernel (own runs, --kernel-trace only, as gpurun requires).
# usage: tools/gpu_pmc.sh <outdir> <stop_after list, e.g. 0 or 2>
set -u
out="$1"; stops="${2:-0}"; corpus="${3:-english}"
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
rm -rf "$R/gpurun_out/$out"; mkdir -p "$R/gpurun_out/$out"   # never mix the CSVs of two runs
run() { # name, counters...
  name="$1"; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$R/gpurun_out/$out/$name" -- python "$R/tools/gpu_ablate.py" "$corpus" 256 "$stops" > "$R/gpurun_out/$out/$name.log" 2>&1
}
run insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT
run cycles SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU
find "$R/gpurun_out/$out" -name "*.csv" | head -20
