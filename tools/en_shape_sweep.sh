#!/bin/bash
# Development aid (GPU box): the seed kernel at several launch shapes (MC_EN_SHAPE=<waves per workgroup>,<workgroups per CU>):
# step time and seed-kernel time per 1 M reads:  tools/en_shape_sweep.sh [read-len]
# (round 5, 150 bp: 8 waves per CU 8.39 ms, 12: 6.44 - 6.46, 16: 6.10 - 6.11, 20: 6.14, 24: 6.21 - 6.29)
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for sh in "8,2" "4,4" "4,5" "4,6" "12,1"; do
echo "shape $sh"; MC_EN_SHAPE=$sh python3 bench.py --steps 4 --warmup 3 --batch 1000000 --resident-batches 2 --read-len ${1:-150} --no-cpu-baseline --no-ags-check --e2e-reads 0 --c5-reads 0 --no-reference-pattern --no-best-only-leg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['config']['kernel_ms_per_step']['k_enumerate'])"
done
