#!/bin/bash
# Calibrates FETCH_SIZE for gathers (GPU box): tools/fetch_calib.sh -> gpurun_out/fetch_calib/calibration.json
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/fetch_calib
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib $R/tools/fetch_calib.hip || exit 1
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o c --output-format csv -- /tmp/fetch_calib > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc1 -o c --output-format csv -- /tmp/fetch_calib > $OUT/pmc1.log 2>&1
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc2 -o c --output-format csv -- /tmp/fetch_calib > $OUT/pmc2.log 2>&1
python3 - $OUT <<'PY'
import collections, csv, glob, json, os, sys
out = sys.argv[1]
req = {"k_stream16": (4 << 30) // 16, "k_gather<32>": 64 << 20, "k_gather<16>": 64 << 20, "k_gather<4>": 64 << 20}
width = {"k_stream16": 16, "k_gather<32>": 32, "k_gather<16>": 16, "k_gather<4>": 4}
def short(n):
    n = n.split("(")[0].replace("void ", "")
    return n
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = {}
for f in glob.glob(os.path.join(out, "trace", "*kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        dur[short(r["Name"])] = float(r["AverageNs"])
doc = {}
for k in req:
    c = {n: sum(v) / len(v) for n, v in agg.get(k, {}).items()}
    d = {"requests": req[k], "bytes_per_request": width[k], "requested_bytes": req[k] * width[k], "avg_ns": dur.get(k), "counters_per_launch": c}
    if "FETCH_SIZE" in c:
        d["FETCH_SIZE_bytes_per_request"] = c["FETCH_SIZE"] * 1024.0 / req[k]       # rocprofv3 reports KiB
        d["FETCH_SIZE_over_requested"] = c["FETCH_SIZE"] * 1024.0 / (req[k] * width[k])
    if dur.get(k):
        d["requested_GBps"] = req[k] * width[k] / dur[k]
    doc[k] = d
json.dump(doc, open(os.path.join(out, "calibration.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(doc, indent=1, sort_keys=True))
PY
find $OUT -name "*.db" -delete
