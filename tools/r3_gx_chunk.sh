#!/bin/bash
# generic engine: bench lines of four workloads (TD_HIP_LIB may name an A/B build, e.g. -DTD_GX_CHUNK=1024)
for cfg in "256 " "64 --single-document" "64 " "8 " "1 "; do
  set -- $cfg
  timeout 300 python bench.py --pattern generic:autogen --size-mb $1 $2 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | \
    python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1 MiB', '$2', j['value'], 'GB/s', j['ms_per_step'], 'ms', j['config']['verified_vs_oracle'], 'generic split', j['roofline']['all_kernels_ms_avg']['td_split_tiles'])"
done
