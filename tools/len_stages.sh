#!/bin/bash
# Development aid (GPU box): the stage times per step for a read length (1 M reads per step): tools/len_stages.sh 300
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
L=${1:-300}
timeout 600 python3 bench.py --steps 6 --warmup 3 --batch ${2:-1000000} --resident-batches 3 --read-len $L --no-cpu-baseline --no-ags-check --e2e-reads 0 --c5-reads 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
c=d['classification_only']
print('L=$L rows %.2f M reads/s %6.2f ms  best only %.2f M reads/s %6.2f ms' % (d['value']/1e6, d['ms_per_step'], c['value']/1e6, c['ms_per_step']))
print(' rows ', d['config']['kernel_ms_per_step']); print(' best ', c['kernel_ms_per_step'])
print(' per read: hsps %.1f gapped %.2f seed hits %.1f rows %.2f' % (d['config']['hsps_per_read'], d['config']['gapped_extensions_per_read'], d['config']['seed_hits_per_read'], d['config']['rows_per_read']))
"
