"""The BASELINE configs[4]-shaped FASTQ file used at size by the sampler tests: 150,000 records of 300 bp sampled from the 30
genomes, phred+33 qualities ~ N(34, 6) clipped to [20, 41], 5 % of the records with one base of quality 10, 2 % exact and 1 %
reverse-complement duplicates of earlier records.  A pure function of its arguments (numpy's frozen RandomState, the seeded read
generator of microbecensus_amd.synth): the golden generator (make_sampler_at_size_golden.py, where /root/reference exists) and
the GPU test write byte-identical files."""
import numpy as np


def write_fastq(path, n=150_000, L=300):
    from microbecensus_amd import synth
    gen = synth.GenomeReads(device="cpu", seed=11)
    r = gen.single(n, L).numpy()
    rng = np.random.RandomState(3)
    comp = bytes.maketrans(b"ACGTN", b"TGCAN")
    qual = (np.clip(np.rint(rng.normal(34, 6, size=(n, L))), 20, 41).astype(np.uint8) + 33)
    low = rng.rand(n) < 0.05
    qual[low, rng.randint(0, L, size=int(low.sum()))] = 33 + 10
    u = rng.rand(n)
    recs, pool = [], []
    for i in range(n):
        sq = bytes(r[i])
        if b"Y" in sq or b"S" in sq:                              # (the genomes hold two IUPAC letters; reverse_complement knows ACGTN only)
            sq = sq.replace(b"Y", b"N").replace(b"S", b"N")
        if pool and u[i] < 0.02:
            sq = pool[rng.randint(len(pool))]
        elif pool and u[i] < 0.03:
            sq = pool[rng.randint(len(pool))][::-1].translate(comp)
        elif len(pool) < 5000:
            pool.append(sq)
        recs.append(b"@s%d\n%s\n+\n%s\n" % (i, sq, bytes(qual[i])))
    with open(path, "wb") as f:
        f.write(b"".join(recs))
    return n


def write_fastq_chunk(path, chunk, n=150_000, L=300):
    """Chunk `chunk` of the FULL-SIZE configs[4] file (tests/test_gpu_at_size.py: 14 chunks = 2.1 M records): the same recipe as
    write_fastq with the chunk's own reads (first = chunk * n), random stream (RandomState(3 + chunk)) and record names; chunk 0 is
    NOT the file of write_fastq (the names differ) - nothing golden depends on this one, its checks are differential."""
    from microbecensus_amd import synth
    gen = synth.GenomeReads(device="cpu", seed=11)
    r = gen.single(n, L, first=chunk * n).numpy()
    rng = np.random.RandomState(3 + chunk)
    comp = bytes.maketrans(b"ACGTN", b"TGCAN")
    qual = (np.clip(np.rint(rng.normal(34, 6, size=(n, L))), 20, 41).astype(np.uint8) + 33)
    low = rng.rand(n) < 0.05
    qual[low, rng.randint(0, L, size=int(low.sum()))] = 33 + 10
    u = rng.rand(n)
    recs, pool = [], []
    for i in range(n):
        sq = bytes(r[i])
        if b"Y" in sq or b"S" in sq:
            sq = sq.replace(b"Y", b"N").replace(b"S", b"N")
        if pool and u[i] < 0.02:
            sq = pool[rng.randint(len(pool))]
        elif pool and u[i] < 0.03:
            sq = pool[rng.randint(len(pool))][::-1].translate(comp)
        elif len(pool) < 5000:
            pool.append(sq)
        recs.append(b"@c%d_%d\n%s\n+\n%s\n" % (chunk, i, sq, bytes(qual[i])))
    with open(path, "wb") as f:
        f.write(b"".join(recs))
    return n


if __name__ == "__main__":       # python c5_at_size.py <repo> <out> <chunk> [n [L]]: one chunk, as a process of its own (the GPU tests start several side by side)
    import sys
    sys.path.insert(0, sys.argv[1])
    write_fastq_chunk(sys.argv[2], int(sys.argv[3]), *(int(a) for a in sys.argv[4:6]))
