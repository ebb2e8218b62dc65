"""Development aid: does running two handles (two half-batches on two host threads / streams) on one GPU overlap the
latency-bound kernels?  python tools/two_engines.py [reads_per_step]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from microbecensus_amd import _native, synth
names, seqs = _native.load_markers(); model = _native.load_model(); fams = model["families"]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
genome = synth.build_genomes(seqs, total_bp=8_000_000, seed=20261001)
reads = bench.sample_reads_device(genome, n, 150, seed=1000, device=torch.device("cuda", 0)); torch.cuda.synchronize()
for ne in (1, 2, 3):
    engs = [_native.Engine(device=0) for _ in range(ne)]
    per = n // ne
    for i, e in enumerate(engs):
        e.set_run(150, model["pars"]["150"], fams); e.attach(reads.data_ptr() + i * per * 150, per)
    def work(e, i, reps):
        for _ in range(reps):
            e.run_range(0, per, first_read_id=i * per)
    for reps in (1, 5):
        th = [threading.Thread(target=work, args=(e, i, reps)) for i, e in enumerate(engs)]
        t0 = time.time(); [t.start() for t in th]; [t.join() for t in th]; dt = time.time() - t0
    print("engines %d: %.1f ms per step of %d reads -> %.2f M reads/s" % (ne, dt / 5 * 1e3, n, n * 5 / dt / 1e6))
    for e in engs: e.close()
