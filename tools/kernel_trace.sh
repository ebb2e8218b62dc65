cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/kt && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --batch 1000000 --resident-batches 2 --no-cpu-baseline --no-ags-check > /tmp/kt.log 2>&1
python3 - <<'PY'
import csv,glob
for f in glob.glob('/tmp/kt/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:14]:
        print(r['Name'][:50].ljust(52), r['Calls'].rjust(4), "%10.3f ms avg" % (float(r['AverageNs'])/1e6), r['Percentage'])
PY
