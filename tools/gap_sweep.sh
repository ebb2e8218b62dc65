#!/bin/bash
# Development aid (GPU box): gapped-extension kernel times for several refill thresholds / occupancies: tools/gap_sweep.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
IFS=";" read -ra CFGS <<< "${GAP_CFGS:-8 5}"
for L in 150 300; do
  for cfg in "${CFGS[@]}"; do
    set -- $cfg
    export MC_GAP_REFILL=$1 MC_GAP_WPC=$2
    OUT=$R/gpurun_out/gapsweep/L${L}_r$1_w$2
    rm -rf $OUT && mkdir -p $OUT
    timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- python3 $R/bench.py --steps 4 --warmup 6 --batch 1000000 --resident-batches 2 --read-len $L --no-cpu-baseline --no-ags-check --e2e-reads 0 --c5-reads 0 > $OUT/trace.log 2>&1
    echo "== L=$L refill=$1 wpc=$2"
    python3 - $OUT <<'PY'
import csv, glob, os, sys
for f in glob.glob(os.path.join(sys.argv[1], "trace", "*kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        n = r["Name"].split("(")[0].replace("void ", "")
        if n.startswith(("k_gap", "k_gapped")):
            print("   %-40s %5s x %10.3f ms" % (n[:40], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
  done
done
