#!/bin/bash
# Development aid (GPU box): a timeline of the kernels of two ranges in flight (mc_set_pipeline): which queue, when, how long.
#   tools/pipeline_trace.sh [read-len] [tail-cus]
L=${1:-150}
CUS=${2:-64}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/ptrace
rm -rf $OUT && mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace -d $OUT/trace -o t --output-format csv -- python3 $R/bench.py --steps 4 --warmup 2 --batch 2000000 --resident-batches 2 --read-len $L --pipeline $CUS --no-serial-leg --no-best-only-leg --no-cpu-baseline --no-ags-check --e2e-reads 0 > $OUT/trace.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "trace", "*kernel_trace.csv"))[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:34], r.get("Queue_Id", ""), r.get("Stream_Id", "")) for r in csv.DictReader(open(f))]
rows.sort()
tr = [i for i, r in enumerate(rows) if r[2].startswith("k_translate_seg")]
a, b = tr[-3], tr[-1]                     # two steps in the steady state
t0 = rows[a][0]
for s, e, n, q, st in rows[a:b]:
    if e - s < 30000: continue            # (30 us: the long ones only)
    print("%8.3f .. %8.3f  %7.3f ms  q%-3s s%-3s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, st, n))
PY
tail -1 $OUT/trace.log | cut -c1-300
