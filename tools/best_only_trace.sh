#!/bin/bash
# Development aid (GPU box): per-kernel durations of the best-hits-only path (tools/best_only_timing.py under rocprofv3): tools/best_only_trace.sh [read-len]
L=${1:-150}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp MC_BOT_ONLY=1
OUT=$R/gpurun_out/botrace_L$L
rm -rf $OUT && mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- python3 $R/tools/best_only_timing.py $L > $OUT/trace.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, os, sys
for f in glob.glob(os.path.join(sys.argv[1], "trace", "*kernel_stats.csv")):
    for r in list(csv.DictReader(open(f)))[:30]:
        print("%-64s %5s x %10.3f ms  %5s %%" % (r["Name"].split("(")[0].replace("void ", "")[:64], r["Calls"], float(r["AverageNs"]) / 1e6, r["Percentage"]))
PY
grep "parts" $OUT/trace.log
