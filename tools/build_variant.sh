#!/bin/bash
# Kernel tuning aid: build the HIP library with extra -D flags into gpurun_out/variants/<name>.so
# usage: tools/build_variant.sh <name> [-DFOO=1 ...]   (run with TD_HIP_LIB=gpurun_out/variants/<name>.so)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p $R/variants
C=$R/tokendagger_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function "$@" \
  $C/td_kernels.hip $C/td_generic.hip $C/td_special.hip $C/td_api.cpp $C/td_tables.cpp $C/td_regex.cpp $C/td_vocab.cpp $C/td_comm.cpp -I$R/include -ldl -o $R/variants/$name.so
echo built $R/variants/$name.so
