#!/bin/bash
# Development aid (GPU box): the finishing kernels after a change - parity first, then the kernel durations at 150 bp (rows and
# best hits only).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | tail -3
[ "${PIPESTATUS[0]}" = 0 ] || exit 1
bash tools/kernel_trace.sh 150 2>&1 | grep -E "k_finish|k_heavy|k_emit|k_gather|k_heads|rocprim|metric" | cut -c1-120
timeout 300 python tools/best_only_timing.py 150 2>&1 | grep "parts 1"
