#!/bin/bash
# Profiles the bench command on the GPU box for ONE read length: tools/profile_round.sh r03 150
#   pass 1: rocprofv3 --kernel-trace --stats            -> per-kernel durations
#   pass 2..: rocprofv3 --pmc (counters only, own runs)  -> SQ / TCC / GRBM / FETCH_SIZE / WRITE_SIZE per kernel
# Raw output goes to gpurun_out/prof_L<len> (scratch); tools/profile_summary.py condenses it into gpurun_out/prof_L<len>/summary/
# as <tag>_L<len>_*, which is what gets copied into profiles/.  One kernel at a time (mc_run_range's default: one part).
TAG=${1:-r03}
L=${2:-150}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_L$L
rm -rf $OUT && mkdir -p $OUT
BENCH="python3 $R/bench.py --steps 4 --warmup 6 --batch 1000000 --resident-batches 2 --read-len $L --no-cpu-baseline --no-ags-check --no-reference-pattern --e2e-reads 0 --c5-reads 0 --no-best-only-leg"
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o $TAG --output-format csv -- $BENCH > $OUT/trace.log 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $set -d $OUT/pmc$i -o $TAG --output-format csv -- $BENCH > $OUT/pmc$i.log 2>&1
done
grep -h '^{' $OUT/trace.log | tail -1 > $OUT/bench_line.json
python3 $R/tools/profile_summary.py $OUT ${TAG}_L$L
find $OUT -name "*.db" -delete
ls -la $OUT/summary
