#!/bin/bash
# Per-kernel durations of a short bench run (development aid, run on the GPU box): tools/kernel_trace.sh [read-len] [extra bench args]
L=${1:-150}; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/ktrace_L$L
rm -rf $OUT && mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- python3 $R/bench.py --steps 4 --warmup 6 --batch 1000000 --resident-batches 2 --read-len $L --no-cpu-baseline --no-ags-check --no-reference-pattern --e2e-reads 0 --c5-reads 0 "$@" > $OUT/trace.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, os, sys
for f in glob.glob(os.path.join(sys.argv[1], "trace", "*kernel_stats.csv")):
    for r in list(csv.DictReader(open(f)))[:28]:
        print("%-64s %5s x %10.3f ms  %5s %%" % (r["Name"].split("(")[0].replace("void ", "")[:64], r["Calls"], float(r["AverageNs"]) / 1e6, r["Percentage"]))
PY
tail -1 $OUT/trace.log | cut -c1-400
