#!/usr/bin/env python3
"""Build the *reference* side of the oracle into oracle/_ref/  (TEST INFRASTRUCTURE ONLY).

The reference's search engine is the bundled, closed RAPsearch2 v2.15 ELF
(/root/reference/microbe_census/bin/rapsearch_Linux_2.15, launched from
microbe_census.py:375) and its marker database `data/rapdb_2.15`, which is MISSING from the
reference tree (.MISSING_LARGE_BLOBS).  There is no source to compile, so the "reference build"
is:

  1. the canonical marker FASTA: training/input/gene_fams/*.faa.gz in sorted() filename order,
     records in file order, first occurrence of each distinct SEQUENCE kept;
  2. `prerapsearch_Linux_2.15 -d markers.dedup.faa -n rapdb_2.15` -> rapdb_2.15 + rapdb_2.15.info;
     the regenerated .info must be byte-identical to the one the reference ships
     (microbe_census/data/rapdb_2.15.info) - that pins the DB *content*;
  3. copies of the two ELF binaries so the oracle can be executed on the GPU box
     (oracle/_ref/ is git-ignored but travels with gpurun).

Nothing under oracle/_ref/ is ever imported, linked or executed by the product
(microbecensus_amd/); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
Run:  python oracle/build_ref.py          (no-op when /root/reference is absent)
"""
import glob
import gzip
import hashlib
import os
import shutil
import subprocess
import sys

REF = os.environ.get("MC_REFERENCE_ROOT", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")

EXPECT_FAA_MD5 = "84a00dcf247a2cf9cbb47d383a675d42"
EXPECT_DB_MD5 = "c31b221b44e37cd074b3c6c66609c29f"


def md5(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def canonical_markers():
    """Yield (id, header_rest, seq) in canonical order, de-duplicated by sequence."""
    seen = set()
    for path in sorted(glob.glob(os.path.join(REF, "training/input/gene_fams/*.faa.gz"))):
        name, chunks = None, []
        with gzip.open(path, "rt") as f:
            for line in f:
                if line.startswith(">"):
                    if name is not None:
                        seq = "".join(chunks)
                        if seq not in seen:
                            seen.add(seq)
                            yield name, seq
                    name, chunks = line[1:].split()[0], []
                else:
                    chunks.append(line.strip())
        if name is not None:
            seq = "".join(chunks)
            if seq not in seen:
                seen.add(seq)
                yield name, seq


def main():
    if not os.path.isdir(REF):
        print("build_ref: %s absent - keeping prebuilt oracle/_ref as is" % REF)
        return 0
    os.makedirs(OUT, exist_ok=True)
    faa = os.path.join(OUT, "markers.dedup.faa")
    db = os.path.join(OUT, "rapdb_2.15")
    if os.path.isfile(db) and os.path.isfile(faa) and md5(db) == EXPECT_DB_MD5:
        print("build_ref: oracle/_ref up to date")
        return 0
    n = 0
    with open(faa, "w") as out:
        for name, seq in canonical_markers():
            out.write(">%s\n%s\n" % (name, seq))
            n += 1
    print("build_ref: %d marker sequences, faa md5 %s" % (n, md5(faa)))
    for b in ("rapsearch_Linux_2.15", "prerapsearch_Linux_2.15"):
        dst = os.path.join(OUT, b)
        shutil.copyfile(os.path.join(REF, "microbe_census/bin", b), dst)
        os.chmod(dst, 0o755)
    subprocess.check_call([os.path.join(OUT, "prerapsearch_Linux_2.15"), "-d", faa, "-n", db],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    shipped = os.path.join(REF, "microbe_census/data/rapdb_2.15.info")
    same = open(shipped, "rb").read() == open(db + ".info", "rb").read()
    print("build_ref: rapdb_2.15 md5 %s ; .info identical to shipped: %s" % (md5(db), same))
    if not same:
        print("build_ref: ERROR regenerated .info differs from the reference's", file=sys.stderr)
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
