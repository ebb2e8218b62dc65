"""Development aid (GPU box): sampling rate and file -> AGS of a FASTQ.bz2 with the block-parallel decoder (csrc/mc_pbzip2.h) against the
one-stream decoder (MC_READER_SERIAL_BZ2).  python tools/bz2_probe.py [nreads]"""
import bz2, contextlib, io, os, subprocess, sys, tempfile, time
from concurrent.futures import ProcessPoolExecutor
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench
import microbecensus_amd; microbecensus_amd.configure_process_env()
from microbecensus_amd import _native, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
gen = synth.GenomeReads(device="cpu", seed=20261001)
td = tempfile.mkdtemp(prefix="mc_bz2_")
plain = os.path.join(td, "reads.fq")
bench.write_fastq(gen, n, 150, plain, False)
text = open(plain, "rb").read()
# one stream per 32 MB of text, compressed side by side (what pbzip2 writes) and - the first 64 MB - as ONE stream (what bzip2 writes)
cuts = [0]
while cuts[-1] < len(text):
    e = min(len(text), cuts[-1] + (32 << 20))
    if e < len(text):
        e = text.rfind(b"\n@", 0, e) + 1
    cuts.append(e)
with ProcessPoolExecutor(8) as ex:
    parts = list(ex.map(bz2.compress, [text[a:b] for a, b in zip(cuts[:-1], cuts[1:])]))
multi = os.path.join(td, "multi.fq.bz2"); open(multi, "wb").write(b"".join(parts))
k = text.rfind(b"\n@", 0, 64 << 20) + 1
single = os.path.join(td, "single.fq.bz2"); open(single, "wb").write(bz2.compress(text[:k]))
for path, nrec in ((multi, n), (single, text[:k].count(b"\n+\n"))):
    for serial in (False, True):
        if serial: os.environ["MC_READER_SERIAL_BZ2"] = "1"
        else: os.environ.pop("MC_READER_SERIAL_BZ2", None)
        t = time.time(); r, st = _native.sample_reads([path], 150, 10**9, True, 33, -5, -5, 100, False); dt = time.time() - t
        print("%s %s: %d records in %.2f s = %.3f M reads/s (%.1f MB of .bz2)" % (os.path.basename(path), "one-stream decoder" if serial else "block-parallel", st["records"], dt, st["records"] / dt / 1e6, os.path.getsize(path) / 1e6), flush=True)
os.environ.pop("MC_READER_SERIAL_BZ2", None)
from microbecensus_amd import microbe_census as mc
for rep in range(2):
    t = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        res = mc.run_pipeline({"seqfiles": [multi], "device": 0, "nreads": n, "read_length": 150})
    dt = time.time() - t
    print("run_pipeline(multi.fq.bz2) run %d: %.2f s = %.2f M reads/s, AGS %.1f" % (rep, dt, n / dt / 1e6, res[0]))
