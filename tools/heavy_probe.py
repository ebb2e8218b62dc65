"""Development aid (GPU box): how heavy are the reads the wave-per-read finishing kernels get?  Rows per read of the bench workload."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import microbecensus_amd; microbecensus_amd.configure_process_env()
from microbecensus_amd import _native, synth
L = int(sys.argv[1]) if len(sys.argv) > 1 else 150
n = 1_000_000
reads = synth.GenomeReads(device="cpu", seed=20261001).single(n, L).numpy()
model = _native.load_model()
eng = _native.Engine(device=0); eng.set_run(L, model["pars"][str(L)], model["families"])
rows, best = eng.search(reads)
st = eng.stats()
per = np.bincount(rows["query"], minlength=n)
print("reads %d, with rows %d, rows %d, hsps %d" % (n, (per > 0).sum(), len(rows), st["hsps"]))
for lo, hi in ((1, 1), (2, 4), (5, 16), (17, 48), (49, 96), (97, 200), (201, 400), (401, 499), (500, 500)):
    m = (per >= lo) & (per <= hi)
    print("rows %3d..%3d: %7d reads, %9d rows" % (lo, hi, m.sum(), per[m].sum()))
# ties among the printed log E of a read's rows (MergeRes sorts by them): reads whose printed keys are all distinct need no replay
q = rows["query"]; le = np.round(rows["loge"], 2)
order = np.lexsort((le, q)); qs, ls = q[order], le[order]
tie = (qs[1:] == qs[:-1]) & (ls[1:] == ls[:-1])
tied_reads = np.unique(qs[1:][tie])
multi = (per >= 2).sum()
print("reads with >= 2 rows: %d; of them with a tie among the printed log E (2 decimals assumed): %d" % (multi, len(tied_reads)))
print("ms:", {k: round(st[k], 3) for k in ("ms_sort", "ms_finish", "ms_total")})
