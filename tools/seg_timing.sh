#!/bin/bash
# Development aid (GPU box): k_translate_seg after a kernel change - parity first, then its duration at 150 and 300 bp, then the
# cycle counters of the timing build (microbecensus_amd/libmc_timing.so, built with -DMC_EXP_TIMING) when it is there.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | tail -3
[ "${PIPESTATUS[0]}" = 0 ] || exit 1
for L in 150 300; do bash tools/kernel_trace.sh $L 2>&1 | grep -E "translate_seg" | cut -c1-120; done
if [ -f microbecensus_amd/libmc_timing.so ]; then
    MCENSUS_LIB=$R/microbecensus_amd/libmc_timing.so timeout 100 python tools/step_timing.py 2>&1 | grep -E "ts-timing|Error" | tail -11 | cut -c1-26,70-
fi
