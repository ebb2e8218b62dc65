p() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['kernel_ms_per_step'])"; }
for v in "$@"; do echo "== $v"; MCENSUS_LIB=$PWD/exp_libs/lib_$v.so python bench.py --steps 2 --warmup 1 --batch 1000000 --resident-batches 1 --no-cpu-baseline --no-ags-check 2>&1 | tail -1 | p; done
echo "== base, counting on"; python bench.py --steps 2 --warmup 1 --batch 1000000 --resident-batches 1 --no-cpu-baseline --no-ags-check --count-in-timed-steps 2>&1 | tail -3 | cut -c1-400
