#!/bin/bash
# Development aid (GPU box): the timed steps one at a time (0), with two ranges in flight on ordinary streams (-1) and for several
# CU splits (mc_set_pipeline) - DESIGN.md 5.5.
#   tools/pipeline_sweep.sh [read-len] [batch]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
L=${1:-150}
B=${2:-2000000}
cd $R
for cus in 0 -1 32 64 128; do
  timeout 600 python3 bench.py --steps 8 --warmup 3 --batch $B --resident-batches 3 --read-len $L --pipeline $cus --no-serial-leg --no-cpu-baseline --no-ags-check --e2e-reads 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
c=d['classification_only']
print('tail CUs %3d  rows %.2f M reads/s (%6.2f ms/step)   best only %.2f M reads/s (%6.2f ms/step)' % ($cus, d['value']/1e6, d['ms_per_step'], c['value']/1e6, c['ms_per_step']), d['config']['kernel_ms_per_step'])
"
done
