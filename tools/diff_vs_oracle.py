"""One-off differential run on the GPU box (development aid): the HIP pipeline's m8 against the oracle's (oracle/rs_port, OpenMP)
on reads of the 30 genomes at several lengths.  python tools/diff_vs_oracle.py [nreads_150]"""
import hashlib, os, subprocess, sys, tempfile, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import microbecensus_amd; microbecensus_amd.configure_process_env()
from microbecensus_amd import _native, synth

n150 = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
gen = synth.GenomeReads(device="cpu", seed=4242)
port, db = os.path.join(REPO, "oracle", "rs_port"), os.path.join(REPO, "oracle", "_ref", "rapdb_2.15")
eng = _native.Engine(device=0)
bad = 0
with tempfile.TemporaryDirectory() as td:
    for L, n in [(int(x.split(":")[0]), int(x.split(":")[1])) for x in os.environ["MC_DIFF_CASES"].split(",")] if os.environ.get("MC_DIFF_CASES") else ((150, n150), (100, n150 // 2), (300, n150 // 4), (50, n150 // 4), (500, n150 // 10)):
        reads = gen.single(n, L, first=L * 1000003).numpy()
        fa = os.path.join(td, "r.fa")
        with open(fa, "w") as f:
            f.write("".join(">%d\n%s\n" % (i, bytes(r).decode()) for i, r in enumerate(reads)))
        eng.set_run(L)
        t = time.time(); rows, _ = eng.search(reads); tg = time.time() - t
        eng.write_m8(os.path.join(td, "gpu.m8"))
        t = time.time(); subprocess.check_call([port, db, fa, os.path.join(td, "cpu.m8")]); tc = time.time() - t
        a = hashlib.md5(open(os.path.join(td, "gpu.m8"), "rb").read()).hexdigest()
        b = hashlib.md5(open(os.path.join(td, "cpu.m8"), "rb").read()).hexdigest()
        print("L=%d n=%d rows=%d gpu %.2fs oracle %.1fs  %s" % (L, n, len(rows), tg, tc, "IDENTICAL" if a == b else "DIFFERENT"), flush=True)
        if a != b:                                              # the first lines that differ, and how many reads are affected
            ga = open(os.path.join(td, "gpu.m8")).read().splitlines(); ca = open(os.path.join(td, "cpu.m8")).read().splitlines()
            sg, sc = set(ga), set(ca)
            og = [x for x in ga if x not in sc]; oc = [x for x in ca if x not in sg]
            print("  lines: gpu %d oracle %d; only gpu %d, only oracle %d; reads affected %d" % (len(ga), len(ca), len(og), len(oc), len(set(x.split("\t")[0] for x in og + oc))))
            for x in og[:6]: print("  gpu   ", x)
            for x in oc[:6]: print("  oracle", x)
            if not og and not oc:
                k = next(i for i in range(min(len(ga), len(ca))) if ga[i] != ca[i])
                print("  same lines, other order from line %d:" % k); print("  gpu   ", ga[k]); print("  oracle", ca[k])
        bad += a != b
sys.exit(1 if bad else 0)
