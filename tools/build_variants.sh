#!/bin/bash
# tools/build_variants.sh <name> <-D flags...>: lightfm_amd/_lib_<name>/liblfm_hip.so = the current objects of lightfm_amd/_lib with
# predict_kernels.hip recompiled with the flags (timing experiments; select with LIGHTFM_AMD_LIB)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
N=$1; shift
mkdir -p $R/lightfm_amd/_lib_$N
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-function "$@" -c $R/lightfm_amd/csrc/predict_kernels.hip -o $R/lightfm_amd/_lib_$N/predict_kernels.o
OBJS=$(ls $R/lightfm_amd/_lib/*.o | grep -v predict_kernels.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/lightfm_amd/_lib_$N/liblfm_hip.so $OBJS $R/lightfm_amd/_lib_$N/predict_kernels.o -ldl
rm -f $R/lightfm_amd/_lib_$N/predict_kernels.o
echo built $R/lightfm_amd/_lib_$N/liblfm_hip.so
