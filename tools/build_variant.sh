#!/bin/bash
# Development aid: an alternative build of the library with extra -D flags, for A/B runs on the GPU box through MCENSUS_LIB
# (exp_libs/ is git-ignored but travels with gpurun):  tools/build_variant.sh <name> -DMC_BIN_LIGHT=64 ...
R=$(cd "$(dirname "$0")/.." && pwd)
N=$1; shift
mkdir -p $R/exp_libs
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wall "$@" -o $R/exp_libs/$N.so $R/microbecensus_amd/csrc/mc_hip.hip $R/microbecensus_amd/csrc/mc_reader.cpp -lz -ldl -pthread
