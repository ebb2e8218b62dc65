#!/bin/bash
# Development aid (GPU box): where the GPU idles between the kernels of consecutive steps (one range at a time): every gap > 20 us
# of two steady-state steps, with the kernels on either side.   tools/gap_trace.sh [read-len]
L=${1:-150}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/gtrace
rm -rf $OUT && mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace --memory-copy-trace -d $OUT/trace -o t --output-format csv -- python3 $R/bench.py --steps 4 --warmup 2 --batch 2000000 --resident-batches 2 --read-len $L --no-best-only-leg --no-cpu-baseline --no-ags-check --e2e-reads 0 --c5-reads 0 > $OUT/trace.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "trace", "*kernel_trace.csv"))[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:34]) for r in csv.DictReader(open(f))]
rows.sort()
tr = [i for i, r in enumerate(rows) if r[2].startswith("k_translate_seg")]
a, b = tr[-3], tr[-1]
sel = rows[a:b + 1]
cur_e, last = sel[0][1], sel[0][2]
idle = 0
for s, e, n in sel[1:]:
    if s > cur_e:
        if s - cur_e > 20000: print("gap %7.3f ms   after %-34s before %s" % ((s - cur_e) / 1e6, last, n))
        idle += s - cur_e
    if e > cur_e: cur_e, last = e, n
print("two steps: span %.3f ms, idle %.3f ms" % ((sel[-1][0] - sel[0][0]) / 1e6, idle / 1e6))
PY
tail -1 $OUT/trace.log | cut -c1-200
