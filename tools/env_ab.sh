#!/bin/bash
# Development aid (GPU box): the stage times of the timed steps with and without an environment switch of the library.
#   tools/env_ab.sh "MC_ORDER_SERIAL=1" ["MC_OTHER=2" ...] [-- read-len batch]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
L=150; B=2000000
run() {
  env $1 timeout 300 python3 bench.py --steps 6 --warmup 3 --batch $B --resident-batches 3 --read-len $L --no-cpu-baseline --no-ags-check --e2e-reads 0 --c5-reads 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
c=d['classification_only']
print('%-28s rows %.2f M reads/s %6.2f ms ' % ('$1', d['value']/1e6, d['ms_per_step']), d['config']['kernel_ms_per_step'])
print('%-28s best %.2f M reads/s %6.2f ms ' % ('', c['value']/1e6, c['ms_per_step']), c['kernel_ms_per_step'])
"
}
run "MC_NONE=0"
for v in "$@"; do run "$v"; done
