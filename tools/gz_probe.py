"""Development aid (GPU box host): sampling rate of a FASTQ.gz with the parallel inflate at several thread counts.
    python3 tools/gz_probe.py [nreads]"""
import os
import sys
import time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import microbecensus_amd; microbecensus_amd.configure_process_env()
from microbecensus_amd import _native, synth
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
L = 150
gen = synth.GenomeReads(device="cuda:0" if torch.cuda.is_available() else "cpu")
path = "/tmp/probe.fq.gz"
t = time.time(); size = bench.write_fastq(gen, n, L, path, True); print("wrote %.2f GB gz in %.1f s, cores %d" % (size / 1e9, time.time() - t, os.cpu_count()))
lib = _native.load_library()
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print(f, open(f).read().strip())
    except Exception as e:
        print(f, "-")
print("affinity", len(os.sched_getaffinity(0)))
os.environ["MC_READER_TIMING"] = "1"
for th in (8, 16, 24, 32):
    if th == 0:
        os.environ["MC_READER_SERIAL_GZ"] = "1"
    else:
        os.environ.pop("MC_READER_SERIAL_GZ", None)
        os.environ["MC_READER_THREADS"] = str(th)
    for rep in range(2):
        t = time.time(); rd = _native.Reader([path], L, n, True, 33, -5, -5, 100, False); k = rd.run(); dt = time.time() - t
        rd.close()
    print("threads %s: %.3f s = %.2f M reads/s" % (th or "serial", dt, k / dt / 1e6))
os.environ["MC_READER_THREADS"] = "32"
t = time.time(); c = _native.count_bases([path]); print("count_bases (32): %.3f s" % (time.time() - t))
