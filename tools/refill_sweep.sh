#!/bin/bash
# Development aid (GPU box): the gapped stage for several refill thresholds (MC_GAP_REFILL) at one read length: tools/refill_sweep.sh 300
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
L=${1:-300}
for rf in 1 2 4 8 16 32; do
  MC_GAP_REFILL=$rf timeout 300 python3 bench.py --steps 4 --warmup 3 --batch 1000000 --resident-batches 2 --read-len $L --no-best-only-leg --no-cpu-baseline --no-ags-check --e2e-reads 0 --c5-reads 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('L=$L refill %2d: gapped stage %.3f ms  (%.2f M reads/s)' % ($rf, d['config']['kernel_ms_per_step']['k_gapped'], d['value']/1e6))
"
done
