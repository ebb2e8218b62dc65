#!/usr/bin/env python3
"""Golden vectors for reads that are NOT clean upper-case ACGT, made by RUNNING THE REFERENCE'S ENGINE here.

The reference hands the sampled reads to `rapsearch_Linux_2.15` as they are (microbe_census.py:352, :375): lower case,
IUPAC codes, `*`, `-`, digits, blanks ... all reach the engine, whose byte tables (`CHashSearch` ctor 0x4169bd-0x416a62,
`BuildQHash@0x40b530`) define what each of the 256 byte values means.  Every golden set captured so far holds clean reads, so
this script takes the reads of BASELINE config 1 that have at least one m8 row (251 of 8,672), makes 12 dirty variants of
each - 1 to 6 random substitutions out of DIRTY plus, in every second variant, a lower-case stretch - and runs the bundled
binary from oracle/_ref on them (`-z 1 -e 1 -t n -p f -b 0`, the reference's own command line).

Only runs where oracle/_ref exists (this container).  Outputs (data only):
  tests/golden/dirty_reads.fa.gz     the 3,012 reads of 100 bytes
  tests/golden/dirty_reads.m8.gz     the engine's m8 (non-# lines)
  tests/golden/dirty_reads.json      counts and md5s
"""
import gzip
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(REPO, "oracle", "_ref")
DIRTY = b"NnacgtRYKMSWBDHVUX*-.0 ?|~"
VARIANTS = 12


def main():
    rng = np.random.RandomState(20261002)
    seqs = [l.strip() for l in gzip.open(os.path.join(HERE, "config1_example_fq.reads.fa.gz"), "rb") if not l.startswith(b">")]
    hit = sorted({int(l.split(b"\t")[0]) for l in gzip.open(os.path.join(HERE, "config1_example_fq.m8.gz"), "rb")})
    out = []
    for q in hit:
        for v in range(VARIANTS):
            s = bytearray(seqs[q])
            for _ in range(rng.randint(1, 7)):
                s[rng.randint(0, len(s))] = DIRTY[rng.randint(0, len(DIRTY))]
            if v & 1:
                a = rng.randint(0, len(s) - 1)
                b = min(len(s), a + rng.randint(1, 30))
                s[a:b] = bytes(s[a:b]).lower()
            if s[-1] in b" \t":                     # (a trailing blank would be stripped with the line end: keep every read 100 bytes long)
                s[-1] = ord("N")
            out.append(bytes(s))
    fasta = b"".join(b">%d\n%s\n" % (i, s) for i, s in enumerate(out))
    with tempfile.TemporaryDirectory() as td:
        fa = os.path.join(td, "dirty.fa")
        open(fa, "wb").write(fasta)
        subprocess.check_call([os.path.join(REF, "rapsearch_Linux_2.15"), "-q", fa, "-d", os.path.join(REF, "rapdb_2.15"), "-o", os.path.join(td, "o"),
                               "-z", "1", "-e", "1", "-t", "n", "-p", "f", "-b", "0"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        m8 = b"".join(l for l in open(os.path.join(td, "o.m8"), "rb") if not l.startswith(b"#"))
    with gzip.GzipFile(os.path.join(HERE, "dirty_reads.fa.gz"), "wb", mtime=0) as f:
        f.write(fasta)
    with gzip.GzipFile(os.path.join(HERE, "dirty_reads.m8.gz"), "wb", mtime=0) as f:
        f.write(m8)
    meta = {"case": "dirty_reads", "source": "config1_example_fq reads with >= 1 m8 row", "variants": VARIANTS, "alphabet": DIRTY.decode(), "reads": len(out),
            "read_length": 100, "m8_rows": m8.count(b"\n"), "m8_md5": hashlib.md5(m8).hexdigest(), "reads_md5": hashlib.md5(fasta).hexdigest(),
            "reads_with_rows": len({l.split(b"\t")[0] for l in m8.splitlines()})}
    json.dump(meta, open(os.path.join(HERE, "dirty_reads.json"), "w"), indent=1, sort_keys=True)
    print(meta)


if __name__ == "__main__":
    sys.exit(main())
