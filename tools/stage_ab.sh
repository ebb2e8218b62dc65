#!/bin/bash
# Development aid (GPU box): stage times per 1 M reads (HIP events of mc_stats) of the product library and of every exp_libs/*.so,
# rows and best hits only:  tools/stage_ab.sh [read-len]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
L=${1:-150}
cd $R
for lib in microbecensus_amd/libmcensus_hip.so exp_libs/*.so; do
  [ -f $lib ] || continue
  echo "== $lib"
  MCENSUS_LIB=$R/$lib timeout 300 python3 bench.py --steps 4 --warmup 3 --batch ${MC_AB_BATCH:-1000000} --resident-batches 2 --read-len $L --no-cpu-baseline --no-ags-check --e2e-reads 0 --c5-reads 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('rows      %.1f M reads/s  %6.2f ms/step ' % (d['value']/1e6, d['ms_per_step']), d['config']['kernel_ms_per_step'])
c=d['classification_only']
print('best only %.1f M reads/s  %6.2f ms/step ' % (c['value']/1e6, c['ms_per_step']), c['kernel_ms_per_step'])
"
done
