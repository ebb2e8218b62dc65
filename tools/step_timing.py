"""Development aid: where does the wall time of one bench step go (kernels / C side / Python side)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import microbecensus_amd; microbecensus_amd.configure_process_env()
from microbecensus_amd import _native, synth
names, seqs = _native.load_markers(); model = _native.load_model(); fams = model["families"]
eng = _native.Engine(device=0); L_ = int(sys.argv[2]) if len(sys.argv) > 2 else 150
eng.set_run(L_, model["pars"][str(L_)], fams)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 150
reads = synth.GenomeReads(device=torch.device("cuda", 0), seed=20261001).single(n, L); torch.cuda.synchronize()
eng.attach(reads.data_ptr(), n)
for it in range(3):
    t0 = time.time(); eng.run_range(0, n); t1 = time.time(); rows, best = eng.rows(copy=False), eng.best_hits(copy=False); t2 = time.time(); st = eng.stats(); t3 = time.time()
    ks = sum(st[k] for k in ("ms_translate", "ms_seed", "ms_eval", "ms_gapped", "ms_sort", "ms_finish"))
    print("run_range %.1f ms (events total %.1f, kernels %.1f) results() %.1f ms stats %.2f ms rows %d best %d hsps %d tasks %d" % ((t1 - t0) * 1e3, st["ms_total"], ks, (t2 - t1) * 1e3, (t3 - t2) * 1e3, len(rows), len(best), st["hsps"], st["seed_tasks"]))
