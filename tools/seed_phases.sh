#!/bin/bash
# Cycle shares of the seed kernel's states (development aid): builds the library with -DMC_EXP_TIMING into exp_libs/ (run this
# part where hipcc is; the file travels with gpurun) and, on a GPU box, prints the per-state cycle counters of one 1 M-read launch.
# tools/seed_phases.sh build | run [read-len]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
if [ "$1" = build ]; then
  mkdir -p $R/exp_libs && cd $R/microbecensus_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -DMC_EXP_TIMING -I../../include -o $R/exp_libs/lib_timing.so mc_hip.hip mc_reader.cpp -lz -ldl -pthread 2>&1 | grep -A3 "error"
  exit 0
fi
L=${2:-150}
MC_PARTS=1 MCENSUS_LIB=$R/exp_libs/lib_timing.so python3 $R/bench.py --steps 1 --warmup 0 --batch 1000000 --resident-batches 1 --read-len $L --no-cpu-baseline --no-ags-check --e2e-reads 0 2>&1 | grep -E "^timing" | tail -6
