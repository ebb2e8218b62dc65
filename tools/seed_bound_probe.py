"""Development aid (GPU box): what binds the seed kernel?  The same 1 M-read launch on (a) 1 M distinct reads of the bench workload and
(b) the first K of them repeated - the same instructions per read, but every index line the launch asks for is in the L2 after the first
repetition.  The difference is what the memory side costs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import microbecensus_amd; microbecensus_amd.configure_process_env()
from microbecensus_amd import _native, synth
L = int(sys.argv[1]) if len(sys.argv) > 1 else 150
n = 1_000_000
gen = synth.GenomeReads(device="cpu", seed=20261001)
reads = gen.single(n, L).numpy()
model = _native.load_model()
eng = _native.Engine(device=0); eng.set_run(L, model["pars"][str(L)], model["families"])
for K in (n, 65536, 4096, 256):
    rr = reads if K == n else np.ascontiguousarray(np.tile(reads[:K], (n // K + 1, 1))[:n])
    eng.upload(rr)
    ms = []
    for it in range(6):
        eng.run_range(0, n)
        st = eng.stats()
        ms.append((st["ms_translate"], st["ms_seed"], st["ms_eval"], st["ms_gapped"], st["ms_sort"], st["ms_finish"]))
    m = np.array(ms[2:]).mean(axis=0)
    print("distinct reads %8d: translate %.2f seed %.2f eval %.2f gapped %.2f sort %.2f finish %.2f ms  (seed hits %d; asks per read: word/exact %.1f wild %.1f pair %.1f probes %.1f)" % (K, *m, st["seed_tasks"], st["seed_exact_asks"] / n, st["seed_wild_asks"] / n, st["seed_pair_asks"] / n, st["seed_probes"] / n), flush=True)
