#!/bin/bash
# How much do the two parts of a range overlap?  Kernel trace of a short bench run with the product's defaults (two parts),
# then per step: wall time of the kernels' union, sum of their durations, idle gaps.  tools/overlap_trace.sh [read-len]
L=${1:-150}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/otrace
rm -rf $OUT && mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace -d $OUT/trace -o t --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --batch 2000000 --resident-batches 2 --read-len $L --no-cpu-baseline --no-ags-check --e2e-reads 0 --c5-reads 0 > $OUT/trace.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "trace", "*kernel_trace.csv"))[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40], r.get("Queue_Id", ""), r.get("Stream_Id", "")) for r in csv.DictReader(open(f))]
rows.sort()
# the timed steps: the last 3 occurrences of k_translate_seg pairs; simply analyse the last 40 % of the trace
t0 = rows[0][0]; t1 = max(r[1] for r in rows)
cut = t0 + (t1 - t0) * 0.55
sel = [r for r in rows if r[0] >= cut]
busy = 0; cur_s, cur_e = sel[0][0], sel[0][1]; gaps = []
for s, e, n, q, st in sel[1:]:
    if s > cur_e: busy += cur_e - cur_s; gaps.append((s - cur_e, n)); cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, *_ in sel)
span = max(r[1] for r in sel) - sel[0][0]
print("span %.2f ms  union-busy %.2f ms  sum-of-kernels %.2f ms  idle %.2f ms  (overlap factor %.2f)" % (span / 1e6, busy / 1e6, tot / 1e6, (span - busy) / 1e6, tot / busy))
gaps.sort(reverse=True)
print("largest idle gaps (ms, next kernel):", [(round(g / 1e6, 3), n) for g, n in gaps[:12]])
print("queues:", sorted(set(r[3] for r in sel)), "streams:", sorted(set(r[4] for r in sel)))
PY
tail -1 $OUT/trace.log | cut -c1-200
