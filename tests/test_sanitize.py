"""The host-side reader (csrc/mc_reader.cpp with mc_pgzip.h and mc_pbzip2.h: parser threads, duplicate-class walkers, inflate and bzip2
workers) under the sanitizers, on the CPU build - the GPU boxes run none.  tests/emul/reader_sanitize.cpp drives every parallel path."""
import bz2
import gzip
import os
import random
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


@pytest.mark.parametrize("san", ["address,undefined", "thread"])
def test_reader_under_sanitizers(san, tmp_path):
    exe = str(tmp_path / "drv")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=" + san, "-fno-omit-frame-pointer", "-o", exe,
                           os.path.join(HERE, "emul", "reader_sanitize.cpp"), os.path.join(REPO, "microbecensus_amd", "csrc", "mc_reader.cpp"), "-lz", "-ldl", "-pthread"])
    rng = random.Random(17)
    comp = str.maketrans("ACGTN", "TGCAN")
    recs, pool = [], []
    for i in range(9000):
        if pool and rng.random() < 0.1:
            s = rng.choice(pool)
            if rng.random() < 0.5:
                s = s[::-1].translate(comp)
        else:
            s = "".join(rng.choice("ACGT") for _ in range(rng.choice([90, 100, 120])))
            pool.append(s)
        lo = rng.choice([2, 25, 30])
        recs.append("@r%d\n%s\n+\n%s\n" % (i, s, "".join(chr(33 + rng.randrange(lo, 41)) for _ in s)))
    text = "".join(recs).encode()
    plain, gz, bz = str(tmp_path / "s.fq"), str(tmp_path / "s.fq.gz"), str(tmp_path / "s.fq.bz2")
    open(plain, "wb").write(text)
    open(gz, "wb").write(gzip.compress(text, 1))
    open(bz, "wb").write(bz2.compress(text[: len(text) // 2], 1) + bz2.compress(text[len(text) // 2:], 1))   # (two streams; the cut is inside a record: the streams' text is one text)
    env = dict(os.environ, MC_READER_THREADS="4", MC_READER_REGION_BYTES="300000", MC_READER_PIECE_BYTES="20000", MC_READER_GZ_CHUNK="65536",
               TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0", ASAN_OPTIONS="detect_leaks=0")
    p = subprocess.run([exe, plain, gz, bz, "90"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    out, err = p.stdout.decode(), p.stderr.decode()
    assert p.returncode == 0, err[-3000:]
    assert "Sanitizer" not in err and "runtime error" not in err, err[-3000:]
    lines = out.splitlines()
    res = {lines[i].strip(): lines[i + 1].strip() for i in range(len(lines) - 1) if lines[i + 1].startswith("  n=")}
    assert res["plain -d"] == res["gz -d"] == res["bz2 -d"] and res["plain"] == res["bz2"], out
    n_d = int(res["plain -d"].split()[0][2:])
    n_all = int(res["plain"].split()[0][2:])
    assert ("describe/walk/take: %d accepted" % n_d) in out and n_d < n_all, out
    assert sum(1 for ln in lines if ln.startswith(("bz2 parts:", "gz parts:")) and ln.endswith("accepted %d" % n_all)) == 2, out
