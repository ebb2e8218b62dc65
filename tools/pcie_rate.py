"""Development aid: PCIe-inclusive rate of mc_search() (host buffer in, rows + best hits back in host memory)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import microbecensus_amd; microbecensus_amd.configure_process_env()
from microbecensus_amd import _native, synth
names, seqs = _native.load_markers(); model = _native.load_model(); fams = model["families"]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
gen = synth.GenomeReads(device="cpu", seed=20261001)          # the bench workload: reads of the reference's 30 genomes
reads = gen.single(n, 150).numpy()
eng = _native.Engine(device=0); eng.set_run(150, model["pars"]["150"], fams)
for it in range(3):
    t = time.time(); eng.lib.mc_search(eng.h, reads.ctypes.data, n, 0); dt = time.time() - t
    st = eng.stats()
    print("mc_search %d reads (pageable host buffer): %.1f ms -> %.2f M reads/s; device %.1f ms; rows %d" % (n, dt * 1e3, n / dt / 1e6, st["ms_total"], st["rows"]))
