#!/bin/bash
# Development aid (GPU box): the timed steps issued as mc_range_end / mc_range_begin / results against mc_run_range per step.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for flag in "--one-at-a-time" ""; do
  timeout 300 python3 bench.py --steps 8 --warmup 3 --batch ${1:-2000000} --resident-batches 3 $flag --no-cpu-baseline --no-ags-check --e2e-reads 0 --c5-reads 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
c=d['classification_only']
print('%-16s rows %.2f M reads/s %6.2f ms  best only %.2f M reads/s %6.2f ms' % ('$flag', d['value']/1e6, d['ms_per_step'], c['value']/1e6, c['ms_per_step']), d['config']['kernel_ms_per_step'])
"
done
