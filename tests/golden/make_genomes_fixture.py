#!/usr/bin/env python3
"""Packs the 30 complete genomes the reference ships for its own read simulator
(/root/reference/training/input/genomes/*.fna.gz, 193 contigs, 84.8 Mbp; SURVEY.md 8(d): the realistic read source
for synthetic benchmarks) into one data fixture: 2 bits per base (A C G T = 0 1 2 3), contig offsets, and the ~100
positions that hold another letter (N, Y, S) with that letter.  Data only - nothing of the reference's code is read.

    python tests/golden/make_genomes_fixture.py        (needs /root/reference; writes tests/golden/genomes/genomes30.npz)
"""
import glob
import gzip
import os

import numpy as np

SRC = "/root/reference/training/input/genomes"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "genomes", "genomes30.npz")


def main():
    code = np.full(256, 255, dtype=np.uint8)
    for i, c in enumerate(b"ACGT"):
        code[c] = i
    seqs, names, genome_of = [], [], []
    for gi, path in enumerate(sorted(glob.glob(os.path.join(SRC, "*.fna.gz")))):
        cur = None
        with gzip.open(path, "rb") as f:
            for line in f:
                if line.startswith(b">"):
                    cur = []
                    seqs.append(cur)
                    names.append(line[1:].split()[0].decode())
                    genome_of.append(gi)
                else:
                    cur.append(line.strip().upper())
    seqs = [np.frombuffer(b"".join(s), dtype=np.uint8) for s in seqs]
    off = np.zeros(len(seqs) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(s) for s in seqs])
    allb = np.concatenate(seqs)
    c = code[allb]
    exc_pos = np.nonzero(c == 255)[0].astype(np.int64)
    exc_chr = allb[exc_pos].copy()
    c[exc_pos] = 0
    pad = (-len(c)) % 4
    c = np.concatenate([c, np.zeros(pad, dtype=np.uint8)]).reshape(-1, 4)
    packed = (c[:, 0] | (c[:, 1] << 2) | (c[:, 2] << 4) | (c[:, 3] << 6)).astype(np.uint8)
    np.savez(OUT, packed=packed, contig_off=off, exc_pos=exc_pos, exc_chr=exc_chr, genome_of=np.array(genome_of, dtype=np.int32),
             names=np.array(names))
    print("%d contigs, %d bases, %d other letters -> %s (%d bytes)" % (len(seqs), off[-1], len(exc_pos), OUT, os.path.getsize(OUT)))


if __name__ == "__main__":
    main()
