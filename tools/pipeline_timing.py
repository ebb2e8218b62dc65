"""Development aid (GPU box): where run_pipeline's wall time goes.  python3 tools/pipeline_timing.py [nreads] [gz]"""
import contextlib
import io
import os
import sys
import time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import microbecensus_amd; microbecensus_amd.configure_process_env()
from microbecensus_amd import _native, synth
from microbecensus_amd import microbe_census as mc
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
gz = len(sys.argv) > 2 and sys.argv[2] == "gz"
L = 150
gen = synth.GenomeReads(device="cuda:0")
path = "/tmp/pt.fq" + (".gz" if gz else "")
bench.write_fastq(gen, n, L, path, gz)
T = {}
def wrap(mod, name):
    f = getattr(mod, name)
    def g(*a, **k):
        t = time.time()
        try:
            return f(*a, **k)
        finally:
            T[name] = T.get(name, 0.0) + time.time() - t
    setattr(mod, name, g)
for nm in ("get_relative_paths", "check_paths", "check_input", "impute_missing_args", "auto_detect_file_type", "auto_detect_quality_offset", "check_arguments", "_sample_search_classify", "classify_reads",
           "aggregate_hits", "clean_up", "estimate_average_genome_size", "_devices_for", "_engines_on"):
    wrap(mc, nm)
orig_sf = _native.Engine.search_files
def sf(self, *a, **k):
    t = time.time()
    try:
        return orig_sf(self, *a, **k)
    finally:
        T["Engine.search_files"] = T.get("Engine.search_files", 0.0) + time.time() - t
_native.Engine.search_files = sf
orig_sr = _native.Engine.set_run
def sr(self, *a, **k):
    t = time.time()
    try:
        return orig_sr(self, *a, **k)
    finally:
        T["Engine.set_run"] = T.get("Engine.set_run", 0.0) + time.time() - t
_native.Engine.set_run = sr
for rep in range(3):
    T.clear()
    args = {"seqfiles": [path], "device": 0, "nreads": n, "read_length": L}
    t = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        res = mc.run_pipeline(args)
    dt = time.time() - t
    print("run %d: %.3f s = %.2f M reads/s | " % (rep, dt, n / dt / 1e6) + "  ".join("%s %.3f" % (k, v) for k, v in sorted(T.items(), key=lambda kv: -kv[1]) if v >= 0.002))
    st = mc._engines[0].stats()
    print("     device ms: total %.1f (translate %.1f seed %.1f eval %.1f gapped %.1f sort %.1f finish %.1f)" % (st["ms_total"], st["ms_translate"], st["ms_seed"], st["ms_eval"], st["ms_gapped"], st["ms_sort"], st["ms_finish"]))
