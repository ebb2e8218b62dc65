#!/bin/bash
# Development aid (GPU box): the wave-per-read finishing kernels with their lists ordered longest first (default) and in the order of
# the atomics (MC_FH_SORT=0): per-kernel durations of a short bench run, alternating:  tools/fh_sort_ab.sh [read-len]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for rep in 1 2; do for v in 1 0; do
  echo "== MC_FH_SORT=$v"
  MC_FH_SORT=$v bash $R/tools/kernel_trace.sh ${1:-150} 2>&1 | grep -E "k_finish_heavy|k_heap_lanes|k_heavy_order" | cut -c1-110
  MC_FH_SORT=$v python3 $R/bench.py --steps 6 --warmup 3 --batch 1000000 --resident-batches 2 --read-len ${1:-150} --no-cpu-baseline --no-ags-check --e2e-reads 0 --c5-reads 0 --no-reference-pattern --no-best-only-leg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('   step %.2f ms  finish %.3f' % (d['ms_per_step'], d['config']['kernel_ms_per_step']['k_finish']))"
done; done
