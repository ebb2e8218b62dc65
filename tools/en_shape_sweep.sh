for sh in "8,2" "12,1" "8,1" "16,1"; do
echo "shape $sh"; MC_EN_SHAPE=$sh python3 bench.py --steps 4 --warmup 3 --batch 1000000 --resident-batches 2 --read-len 150 --no-cpu-baseline --no-ags-check --e2e-reads 0 --no-reference-pattern 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['config']['kernel_ms_per_step'])"
done
