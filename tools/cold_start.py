"""Development aid (GPU box): the cold start of the reference's default use - ONE run_pipeline of 1 - 2 M reads per process
(/root/reference/scripts/run_microbe_census.py:31).  Writes a plain FASTQ of n reads of the bench workload, then runs
scripts/run_microbe_census.py -n <n> on it in fresh processes (wall time incl. the interpreter) and once with MC_OPEN_TIMING.
python tools/cold_start.py [n] [reps]"""
import os, subprocess, sys, tempfile, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
import bench
import microbecensus_amd; microbecensus_amd.configure_process_env()
from microbecensus_amd import synth
gen = synth.GenomeReads(device="cpu", seed=20261001)
td = tempfile.mkdtemp(prefix="mc_cold_")
fq = os.path.join(td, "reads.fq")
bench.write_fastq(gen, n, 150, fq, False)
cli = os.path.join(REPO, "scripts", "run_microbe_census.py")
for rep in range(reps):
    out = os.path.join(td, "out%d.txt" % rep)
    t = time.time()
    subprocess.check_call([sys.executable, cli, "-n", str(n), fq, out], env=dict(os.environ, MC_OPEN_TIMING="1") if rep % 2 else None)
    dt = time.time() - t
    ags = [l for l in open(out) if l.startswith("average_genome_size")][0].strip()
    print("cold CLI run %d: %.3f s  (%s)" % (rep, dt, ags), flush=True)
t = time.time()
subprocess.check_call([sys.executable, "-c", "import numpy"])
print("python -c 'import numpy': %.3f s" % (time.time() - t))
t = time.time()
subprocess.check_call([sys.executable, "-X", "importtime", "-c", "import sys; sys.path.insert(0, %r); from microbecensus_amd import microbe_census" % REPO], stderr=subprocess.DEVNULL)
print("import microbecensus_amd.microbe_census: %.3f s" % (time.time() - t))
env = dict(os.environ, MC_OPEN_TIMING="1")
subprocess.check_call([sys.executable, cli, "-n", str(n), fq, os.path.join(td, "o.txt")], env=env)
