"""Development aid (GPU box): per-stage times of the rows path and of the best-hits-only path on the bench workload.
    python3 tools/best_only_timing.py [read_len]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import microbecensus_amd; microbecensus_amd.configure_process_env()
from microbecensus_amd import _native, synth

L = int(sys.argv[1]) if len(sys.argv) > 1 else 150
dev = torch.device("cuda", 0)
gen = synth.GenomeReads(device=dev, seed=20261001)
N = 2_000_000
reads = gen.single(2 * N, L)
torch.cuda.synchronize()
model = _native.load_model()
eng = _native.Engine(device=0)
eng.set_run(L, model["pars"][str(L)], model["families"])
eng.attach(reads.data_ptr(), 2 * N)
import time
only_modes = (True,) if os.environ.get("MC_BOT_ONLY") else (False, True)      # (tools/best_only_trace.sh: the best-hits-only path alone)
for parts in ((1,) if os.environ.get("MC_BOT_ONLY") else (1, 2)):
    for only in only_modes:
        eng.set_best_hits_only(only)
        for rep in range(3):
            t = time.time()
            eng.run_range((rep % 2) * N, N, first_read_id=(rep % 2) * N)
            dt = time.time() - t
        st = eng.stats()
        print("parts %d best_only %d: %.2f ms per 2 M reads (%.1f M reads/s)  translate %.2f seed %.2f eval %.2f gapped %.2f sort %.2f finish %.2f | hsps %d rows %d classified %d" % (
            parts, only, dt * 1e3, N / dt / 1e6, st["ms_translate"], st["ms_seed"], st["ms_eval"], st["ms_gapped"], st["ms_sort"], st["ms_finish"], st["hsps"], st["rows"], st["classified"]))
