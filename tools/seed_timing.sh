#!/bin/bash
# Development aid (GPU box): the seed kernels after a change - parity first, then the kernel durations at 150 and 300 bp, then the
# cycle counters of the timing build (microbecensus_amd/libmc_timing.so, built with -DMC_EXP_TIMING) when it is there.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | tail -3
[ "${PIPESTATUS[0]}" = 0 ] || exit 1
for L in 150 300; do bash tools/kernel_trace.sh $L 2>&1 | grep -E "k_enumerate|k_eval_seeds|metric" | cut -c1-120; done
if [ -f microbecensus_amd/libmc_timing.so ]; then
    MCENSUS_LIB=$R/microbecensus_amd/libmc_timing.so timeout 100 python tools/step_timing.py 2>&1 | grep -E "^timing|Error" | tail -6
fi
