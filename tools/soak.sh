#!/bin/bash
# Development aid (GPU box): three runs of 30 steps (60 M reads each) - the counts and the AGS of the workload must be the same every time
# (the kernels take their work from counters: which wave does what differs from run to run, the results must not).
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
timeout 600 python3 bench.py --steps 30 --warmup 2 --resident-batches 6 --no-cpu-baseline --no-ags-check --e2e-reads 0 --c5-reads 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']
print('run $i: %.2f M reads/s classified %d rows/read %.6f hsps/read %.4f gapped/read %.4f seeds/read %.3f ags %s' % (d['value']/1e6, c['classified_reads'], c['rows_per_read'], c['hsps_per_read'], c['gapped_extensions_per_read'], c['seed_hits_per_read'], c['ags_estimate_of_workload']))
"
done
