"""Where the end-to-end time goes (development aid, GPU box): python tools/e2e_probe.py [nreads]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
import microbecensus_amd; microbecensus_amd.configure_process_env()
from microbecensus_amd import _native, synth
from microbecensus_amd import microbe_census as mc
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
L = 150
gen = synth.GenomeReads(device="cuda:0")
reads = gen.single(n, L).cpu().numpy()
w = len(str(n - 1))
rec = np.empty((n, 1 + w + 1 + L + 3 + L + 1), dtype=np.uint8)
rec[:, 0] = ord("@"); ids = np.arange(n)
for k in range(w):
    rec[:, w - k] = ord("0") + (ids // 10 ** k) % 10
rec[:, 1 + w] = 10; rec[:, 2 + w:2 + w + L] = reads; rec[:, 2 + w + L:5 + w + L] = np.frombuffer(b"\n+\n", dtype=np.uint8)
rec[:, 5 + w + L:5 + w + 2 * L] = ord("I"); rec[:, 5 + w + L + 7:5 + w + 2 * L:10] = ord("5"); rec[:, -1] = 10
path = "/tmp/probe.fq"
rec.tofile(path)
print("file", os.path.getsize(path) / 1e9, "GB", "cores", os.cpu_count())
for th in (16, 32):
    os.environ["MC_READER_THREADS"] = str(th)
    t = time.time(); rd = _native.Reader([path], L, n, True, 33, -5, -5, 100, False); k = rd.run(); dt = time.time() - t
    print("reader threads %d: %.3f s = %.2f M reads/s" % (th, dt, k / dt / 1e6)); rd.close()
os.environ["MC_READER_THREADS"] = "32"
t = time.time(); c = _native.count_bases([path]); print("count_bases %.3f s" % (time.time() - t), c)
model = _native.load_model(); fams = model["families"]
t = time.time(); eng = _native.Engine(device=0); print("engine open %.3f s" % (time.time() - t))
eng.set_run(L, model["pars"][str(L)], fams)
for th in (32, 8, 10, 12, 14, 16, 20, 24, 32):     # (the box grants 16 CPUs of its 256: more runnable threads than that and the quota throttles ALL of them, the one that drives the GPU too)
    os.environ["MC_READER_THREADS"] = str(th)
    for rep in range(2):
        rd = _native.Reader([path], L, n, True, 33, -5, -5, 100, False)
        t = time.time(); rows, best = eng.search_files(rd, keep_rows=False, best_only=True); dt = time.time() - t
        st = eng.stats()
        print("reader threads %2d: search_files (best hits only) run %d: %.3f s = %.2f M reads/s  (device: ranges %.3f s)" % (th, rep, dt, n / dt / 1e6, st["ms_total"] / 1e3), "sampler run %.3f parse %.3f copies %.3f" % (rd.times()["run"], rd.times()["parse"], rd.times()["verdicts_places_copies"]), "gapped %.1f ms" % st["ms_gapped"]); rd.close()
os.environ["MC_READER_THREADS"] = "32"
for rep in range(2):
    t = time.time(); rows, best = eng.search(reads); dt = time.time() - t
    print("mc_search from host memory run %d: %.3f s = %.2f M reads/s rows %d" % (rep, dt, n / dt / 1e6, len(rows)))
# the same reads resident in HBM, best hits only, ranges of 2 M one after the other (what bench.py times)
rr = gen.single(n, L); torch.cuda.synchronize()
eng.attach(rr.data_ptr(), n); eng.set_best_hits_only(True)
for rep in range(2):
    acc = {}
    t = time.time()
    for lo in range(0, n, 2_000_000):
        eng.run_range(lo, min(2_000_000, n - lo), first_read_id=lo); st = eng.stats()
        for k in ("ms_translate", "ms_seed", "ms_eval", "ms_gapped", "ms_sort", "ms_finish"): acc[k] = acc.get(k, 0) + st[k]
    dt = time.time() - t
    print("resident, best hits only, run %d: %.3f s = %.2f M reads/s" % (rep, dt, n / dt / 1e6), {k: round(v, 1) for k, v in acc.items()})
eng.set_best_hits_only(False); eng.attach(0, 0)
mc._engines[0] = eng
import contextlib, io
for rep in range(2):
    args = {"seqfiles": [path], "device": 0, "nreads": n, "read_length": L}
    t = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        res = mc.run_pipeline(args)
    dt = time.time() - t
    print("run_pipeline run %d: %.3f s = %.2f M reads/s" % (rep, dt, n / dt / 1e6), res[0])
