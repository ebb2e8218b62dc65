"""Development aid (GPU box): file -> AGS on a FASTQ.gz for several splits of the host's CPUs between the inflate workers
(MC_READER_GZ_THREADS) and the record parsers (MC_READER_THREADS).  python tools/gz_threads.py [nreads]"""
import contextlib, io, os, subprocess, sys, tempfile, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench
import microbecensus_amd; microbecensus_amd.configure_process_env()
from microbecensus_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
gen = synth.GenomeReads(device="cpu", seed=20261001)
td = tempfile.mkdtemp(prefix="mc_gz_")
path = os.path.join(td, "reads.fq.gz")
bench.write_fastq(gen, n, 150, path, True)
code = """
import contextlib, io, sys, time
sys.path.insert(0, %r)
from microbecensus_amd import microbe_census as mc
w = []
for rep in range(3):
    t = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        mc.run_pipeline({"seqfiles": [%r], "device": 0, "nreads": %d, "read_length": 150})
    w.append(time.time() - t)
print("%%.3f" %% min(w[1:]))
""" % (REPO, path, n)
for gz, rd in ((0, 0), (12, 8), (12, 4), (12, 3), (13, 3), (14, 2), (11, 5), (10, 6), (12, 6), (0, 0)):
    env = dict(os.environ)
    if gz: env["MC_READER_GZ_THREADS"] = str(gz); env["MC_READER_THREADS"] = str(rd)
    out = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode().strip()
    print("gz workers %2s parsers %2s: %s s = %.2f M reads/s" % (gz or "def", rd or "def", out, n / float(out) / 1e6), flush=True)
