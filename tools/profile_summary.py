"""Condenses the rocprofv3 output of tools/profile_round.sh into small per-kernel tables (what profiles/ keeps)."""
import collections
import csv
import glob
import json
import os
import sys

out, tag = sys.argv[1], sys.argv[2]
dst = os.path.join(out, "summary")
os.makedirs(dst, exist_ok=True)


def short(name):
    name = name.split("(")[0]
    if "rocprim" in name:
        return "rocprim::" + name.split("::")[-1][:40] if "::" in name else name[:60]
    return name.replace("void ", "")[:60]


# kernel stats: keep rocprofv3's own summary (names shortened)
for f in glob.glob(os.path.join(out, "trace", "*kernel_stats.csv")):
    rows = list(csv.DictReader(open(f)))
    with open(os.path.join(dst, "%s_kernel_stats.csv" % tag), "w") as o:
        w = csv.writer(o)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])

# counters: sum over dispatches and divide by the number of dispatches -> per-launch averages
launches = collections.Counter()
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in sorted(glob.glob(os.path.join(out, "pmc*", "*counter_collection.csv"))):
    seen = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        seen[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
    for (k, c), ids in seen.items():
        launches[(k, c)] = len(ids)
names = sorted({c for v in agg.values() for c in v})
with open(os.path.join(dst, "%s_pmc_per_launch.csv" % tag), "w") as o:
    w = csv.writer(o)
    w.writerow(["Kernel", "Launches"] + names)
    for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", 0)):
        if not k.startswith(("k_", "rocprim")):
            continue
        n = max(launches[(k, c)] for c in agg[k])
        w.writerow([k, n] + ["%.6g" % (agg[k][c] / launches[(k, c)]) if c in agg[k] else "" for c in names])
# HBM traffic per launch: FETCH_SIZE / WRITE_SIZE are in KiB... rocprofv3 reports them in kilobytes (1024 B)
traffic = {}
for k in agg:
    if "FETCH_SIZE" in agg[k]:
        traffic[k] = {"fetch_kib_per_launch": agg[k]["FETCH_SIZE"] / launches[(k, "FETCH_SIZE")],
                      "write_kib_per_launch": (agg[k]["WRITE_SIZE"] / launches[(k, "WRITE_SIZE")]) if "WRITE_SIZE" in agg[k] else None}
json.dump(traffic, open(os.path.join(dst, "%s_hbm_traffic.json" % tag), "w"), indent=1, sort_keys=True)
bl = os.path.join(out, "bench_line.json")
if os.path.exists(bl):
    open(os.path.join(dst, "%s_bench_line.json" % tag), "w").write(open(bl).read())
