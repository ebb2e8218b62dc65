#!/bin/bash
# The scattered-line ceiling of the box's GPU (GPU box): tools/gather_ceiling.sh -> gpurun_out/gather_ceiling.json
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/gather_ceiling $R/tools/gather_ceiling.hip || exit 1
timeout 300 /tmp/gather_ceiling > $R/gpurun_out/gather_ceiling.json || exit 1
cat $R/gpurun_out/gather_ceiling.json
