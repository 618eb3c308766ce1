#!/bin/bash
# same-box A/B of library builds: tools/r3_ab_libs.sh "<corpus list>" <lib> [<lib> ...]   (each library twice, interleaved)
mkdir -p gpurun_out
corpora=$1; shift
for c in $corpora; do
  for rep in 1 2; do
    for lib in "$@"; do
      echo "== $c lib=$lib"
      TD_HIP_LIB=$lib TD_AB_MODES=1 timeout 300 python tools/gpu_ab.py $c 256 10 2>&1 | tail -1
    done
  done
done
