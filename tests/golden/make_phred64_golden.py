#!/usr/bin/env python3
"""Golden for a phred+64 FASTQ through the quality filter (VERDICT r05 "missing" #6), produced by RUNNING THE REFERENCE here:
auto_detect_quality_offset (microbe_census.py:175-187) must answer 64 - the first record holds only qualities 0 .. 10, the
characters '@' .. 'J' that decide nothing, so the detection has to walk into the second record -, and quality_filter (:265-279)
must apply -q 10 / -m 25 to ord(c) - 64.  Input: 20,000 reads of 110 bp of the 30-genome fixture (microbecensus_amd/synth.py;
data only), qualities ~ N(30, 8) clipped to [2, 40], a tenth of the reads with a tail of quality 2 ('B', Illumina 1.5's read
segment indicator).  Writes tests/golden/inputs/phred64_110bp.fq.gz and tests/golden/c1_phred64_q_m.{json,m8.gz,reads.fa.gz}.
Only runs where /root/reference exists."""
import gzip
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)
from make_golden import load_reference, run_case   # noqa: E402


def write_input(path, n=20000, L=110):
    import numpy as np
    from microbecensus_amd import synth
    gen = synth.GenomeReads(device="cpu", seed=20261001)
    r = gen.single(n, L, first=50_000_000).numpy()
    rng = np.random.RandomState(64)
    q = np.clip(np.rint(rng.normal(30, 8, size=(n, L))), 2, 40).astype(np.int64)
    tail = rng.rand(n) < 0.10
    for i in np.nonzero(tail)[0]:
        q[i, rng.randint(20, L):] = 2
    q[0] = rng.randint(0, 11, size=L)                       # the first record decides nothing: '@' .. 'J'
    with gzip.GzipFile(path, "wb", mtime=0) as f:
        f.write(b"".join(b"@p%d\n%s\n+\n%s\n" % (i, bytes(r[i]), bytes((q[i] + 64).astype(np.uint8))) for i in range(n)))


def main():
    mc, scratch = load_reference()
    inp = os.path.join(HERE, "inputs", "phred64_110bp.fq.gz")
    if not os.path.exists(inp):
        write_input(inp)
    args = {"seqfiles": [inp], "threads": 8, "read_length": 100, "min_quality": 10, "mean_quality": 25}
    out = run_case(mc, "c1_phred64_q_m", args)
    assert out["args"]["quality_offset"] == 64, out["args"]
    print({k: v for k, v in out["args"].items()})


if __name__ == "__main__":
    main()
