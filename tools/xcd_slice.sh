#!/bin/bash
# Does a scattered table run at the L2's rate when every XCD is only asked for its eighth?  (GPU box): tools/xcd_slice.sh -> gpurun_out/xcd_slice.json
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/xcd_slice $R/tools/xcd_slice.hip || exit 1
timeout 300 /tmp/xcd_slice > $R/gpurun_out/xcd_slice.json || exit 1
cat $R/gpurun_out/xcd_slice.json
