#!/bin/bash
# Resource usage and instruction mix of the HIP kernels (development aid): tools/kernel_report.sh [name-filter]
cd "$(dirname "$0")/../microbecensus_amd/csrc" || exit 1
F=${1:-k_}
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wall -S --cuda-device-only -o /tmp/mc_hip.s mc_hip.hip -Rpass-analysis=kernel-resource-usage 2>&1 \
  | grep -E "error|Function Name" -A9 | grep -E "error|Function Name|VGPRs:|Spill|ScratchSize|Occupancy" | grep -A6 -E "error|Function Name: .*$F" | sed 's/\[-Rpass.*//; s/[a-z_]*\.[hip]*:[0-9]*:[0-9]*: remark: //'
python3 - "$F" <<'PY'
import re, sys
s = open('/tmp/mc_hip.s').read()
for f in re.split(r'\n(?=_Z[\w]+:)', s):
    name = f.split(':')[0]
    if sys.argv[1] in name and name.startswith('_Z'):
        ins = [l.strip() for l in f.split('\n')]
        ins = [l for l in ins if l and not l.startswith(('.', ';', '_Z')) and not l.endswith(':')]
        c = lambda *p: sum(1 for l in ins if l.startswith(p))
        print(name[:60], 'total', len(ins), 'valu', c('v_'), 'salu', c('s_'), 'vmem', c('global_', 'buffer_', 'scratch_'), 'flat', c('flat_'), 'lds', c('ds_'), 'waitcnt', c('s_waitcnt'))
PY
