"""Where run_pipeline's wall time goes beside the native call (development aid, GPU box): python tools/pipeline_stages.py [nreads]"""
import os, sys, time, contextlib, io
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import microbecensus_amd; microbecensus_amd.configure_process_env()
from microbecensus_amd import _native, synth
from microbecensus_amd import microbe_census as mc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
L = 150
gen = synth.GenomeReads(device="cpu")
path = "/tmp/stages.fq"
with open(path, "wb") as f:
    for lo in range(0, n, 1_000_000):
        k = min(1_000_000, n - lo)
        reads = gen.single(k, L, first=lo).numpy()
        w = 8
        rec = np.empty((k, 1 + w + 1 + L + 3 + L + 1), dtype=np.uint8)
        rec[:, 0] = ord("@"); ids = np.arange(lo, lo + k)
        for j in range(w):
            rec[:, w - j] = ord("0") + (ids // 10 ** j) % 10
        rec[:, 1 + w] = 10; rec[:, 2 + w:2 + w + L] = reads; rec[:, 2 + w + L:5 + w + L] = np.frombuffer(b"\n+\n", dtype=np.uint8)
        rec[:, 5 + w + L:5 + w + 2 * L] = ord("I"); rec[:, -1] = 10
        rec.tofile(f)
timers = {}
def wrap(mod, name):
    fn = getattr(mod, name)
    def inner(*a, **k):
        t = time.time()
        try:
            return fn(*a, **k)
        finally:
            timers[name] = timers.get(name, 0.0) + time.time() - t
    setattr(mod, name, inner)
for nm in ("check_input", "impute_missing_args", "check_arguments", "_sample_search_classify", "classify_reads", "aggregate_hits", "estimate_average_genome_size", "clean_up", "get_relative_paths", "check_paths"):
    wrap(mc, nm)
for nm in ("set_run", "search_files"):
    wrap(_native.Engine, nm)
for rep in range(3):
    timers.clear()
    args = {"seqfiles": [path], "device": 0, "nreads": n}
    t = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        res = mc.run_pipeline(args)
    dt = time.time() - t
    print("run %d: %.3f s = %.2f M reads/s" % (rep, dt, n / dt / 1e6), {k: round(v, 3) for k, v in timers.items()})
import cProfile, pstats
args = {"seqfiles": [path], "device": 0, "nreads": n, "read_length": L}
pr = cProfile.Profile()
with contextlib.redirect_stdout(io.StringIO()):
    pr.enable(); res = mc.run_pipeline(args); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
