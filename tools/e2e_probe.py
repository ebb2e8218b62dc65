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
for th in (8, 16, 32, 64):
    os.environ["MC_READER_THREADS"] = str(th)
    t = time.time(); rd = _native.Reader([path], L, n, True, 33, -5, -5, 100, False); k = rd.run(); dt = time.time() - t
    print("reader threads %d: %.3f s = %.2f M reads/s" % (th, dt, k / dt / 1e6)); rd.close()
os.environ["MC_READER_THREADS"] = "32"
t = time.time(); c = _native.count_bases([path]); print("count_bases %.3f s" % (time.time() - t), c)
model = _native.load_model(); fams = model["families"]
t = time.time(); eng = _native.Engine(device=0); print("engine open %.3f s" % (time.time() - t))
eng.set_run(L, model["pars"][str(L)], fams)
for rep in range(3):
    rd = _native.Reader([path], L, n, True, 33, -5, -5, 100, False)
    t = time.time(); rows, best = eng.search_files(rd, keep_rows=False); dt = time.time() - t
    print("search_files run %d: %.3f s = %.2f M reads/s  (device %.3f s)" % (rep, dt, n / dt / 1e6, eng.stats()["ms_total"] / 1e3), len(best)); rd.close()
t = time.time(); rows, best = eng.search(reads); dt = time.time() - t
print("mc_search from host memory: %.3f s = %.2f M reads/s rows %d" % (dt, n / dt / 1e6, len(rows)))
mc._engines[0] = eng
import contextlib, io
for rep in range(2):
    args = {"seqfiles": [path], "device": 0, "nreads": n, "read_length": L}
    t = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        res = mc.run_pipeline(args)
    dt = time.time() - t
    print("run_pipeline run %d: %.3f s = %.2f M reads/s" % (rep, dt, n / dt / 1e6), res[0])
