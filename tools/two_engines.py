"""Development aid: how much would free-running (staggered) pipelines gain over the two in-step parts of one call?  Two or three
handles on one GPU, one host thread each, every one looping over its own resident reads with ONE part per call; aggregate rate
against a single handle with two parts.  python tools/two_engines.py [reads_per_call]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import microbecensus_amd; microbecensus_amd.configure_process_env()
from microbecensus_amd import _native, synth
import torch
model = _native.load_model(); fams = model["families"]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
L = 150
gen = synth.GenomeReads(device="cuda:0")
reps = 12
for ne, parts in ((1, 2), (2, 1), (3, 1), (2, 2)):
    reads = gen.single(n * ne, L).contiguous(); torch.cuda.synchronize()
    engs = [_native.Engine(device=0) for _ in range(ne)]
    for i, e in enumerate(engs):
        e.set_run(L, model["pars"][str(L)], fams); e.lib.mc_set_keep_rows(e.h, 0)
        e.attach(reads.data_ptr() + i * n * L, n)
        e.run_range(0, n, first_read_id=i * n)                     # warm-up: pools
    def work(e, i):
        for _ in range(reps):
            e.run_range(0, n, first_read_id=i * n)
    th = [threading.Thread(target=work, args=(e, i)) for i, e in enumerate(engs)]
    t0 = time.time(); [t.start() for t in th]; [t.join() for t in th]; dt = time.time() - t0
    print("handles %d x parts %d, %d reads per call: %.2f M reads/s" % (ne, parts, n, n * ne * reps / dt / 1e6), flush=True)
    for e in engs: e.close()
