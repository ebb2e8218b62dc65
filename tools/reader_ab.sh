#!/bin/bash
# Development aid (GPU box): the product library against exp_libs/*.so on the file -> best hits path (tools/e2e_probe.py), same box, alternating
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for rep in 1 2; do
for lib in microbecensus_amd/libmcensus_hip.so exp_libs/*.so; do
  [ -f $lib ] || continue
  echo "== $lib"
  MCENSUS_LIB=$R/$lib timeout 600 python3 tools/e2e_probe.py ${1:-20000000} 2>/dev/null | grep -E "reader threads|search_files"
done
done
