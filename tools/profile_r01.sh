set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/trace -o r01 --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --batch 1000000 --resident-batches 2 --no-cpu-baseline > $R/gpurun_out/prof/trace.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA -d $R/gpurun_out/prof/pmc1 -o r01 --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --batch 1000000 --resident-batches 1 --no-cpu-baseline > $R/gpurun_out/prof/pmc1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES -d $R/gpurun_out/prof/pmc2 -o r01 --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --batch 1000000 --resident-batches 1 --no-cpu-baseline > $R/gpurun_out/prof/pmc2.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof/pmc3 -o r01 --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --batch 1000000 --resident-batches 1 --no-cpu-baseline > $R/gpurun_out/prof/pmc3.log 2>&1
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum -d $R/gpurun_out/prof/pmc4 -o r01 --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --batch 1000000 --resident-batches 1 --no-cpu-baseline > $R/gpurun_out/prof/pmc4.log 2>&1
find $R/gpurun_out/prof -type f | head -50; du -sh $R/gpurun_out/prof
find $R/gpurun_out/prof -name "*.db" -delete
for f in $R/gpurun_out/prof/*.log; do echo "== $f"; tail -5 $f | cut -c1-600; done
find $R/gpurun_out/prof -type f | xargs ls -la
