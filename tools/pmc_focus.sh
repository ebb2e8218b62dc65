#!/bin/bash
# Counters of ONE kernel (development aid, run on the GPU box): tools/pmc_focus.sh <kernel-regex> [read-len]
# PMC_SETS="A B;C D" replaces the default counter sets.  Each counter set is its own rocprofv3 --pmc pass (no tracing options beside it); prints per-launch averages.
K=${1:-k_gapped}
L=${2:-150}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/focus
rm -rf $OUT && mkdir -p $OUT
BENCH="python3 $R/bench.py --steps 3 --warmup 6 --batch 1000000 --resident-batches 1 --read-len $L --no-cpu-baseline --no-ags-check --no-reference-pattern --e2e-reads 0 --c5-reads 0"
i=0
if [ -n "$PMC_SETS" ]; then IFS=';' read -ra SETS <<< "$PMC_SETS"; else SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INSTS_BRANCH SQ_INSTS_SMEM" \
           "FETCH_SIZE" "WRITE_SIZE"); fi
for set in "${SETS[@]}"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-include-regex "$K" -d $OUT/pmc$i -o f --output-format csv -- $BENCH > $OUT/pmc$i.log 2>&1
done
python3 - $OUT <<'PY'
import collections, csv, glob, os, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in sorted(glob.glob(os.path.join(sys.argv[1], "pmc*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:50]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
for k in agg:
    print(k)
    for c in sorted(agg[k]):
        print("   %-28s %14.6g  (%d launches)" % (c, agg[k][c] / len(n[(k, c)]), len(n[(k, c)])))
PY
