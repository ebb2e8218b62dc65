"""Deterministic synthetic shotgun reads for benchmarks and size-independent tests.

No reference code or data is involved beyond the marker proteins shipped as package data: a synthetic
"community" of genomes is assembled from (a) marker proteins picked from the database and diverged at the
amino-acid level and (b) random ORFs with the database's amino-acid composition, all back-translated with
random synonymous codons, placed on either strand and separated by random spacers.  About 1 % of the genes
are markers, which is what a bacterial genome carries of the 30 universal families.  Reads are sampled
uniformly over the genomes, strand by fair coin, error free.

The generator is counter based (splitmix64 of seed and index), so every value is a pure function of
(seed, index): identical on every platform and cheap to vectorise with numpy.
"""
import os

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)

CODONS = {
    "A": ["GCT", "GCC", "GCA", "GCG"], "R": ["CGT", "CGC", "CGA", "CGG", "AGA", "AGG"], "N": ["AAT", "AAC"], "D": ["GAT", "GAC"],
    "C": ["TGT", "TGC"], "Q": ["CAA", "CAG"], "E": ["GAA", "GAG"], "G": ["GGT", "GGC", "GGA", "GGG"], "H": ["CAT", "CAC"],
    "I": ["ATT", "ATC", "ATA"], "L": ["TTA", "TTG", "CTT", "CTC", "CTA", "CTG"], "K": ["AAA", "AAG"], "M": ["ATG"], "F": ["TTT", "TTC"],
    "P": ["CCT", "CCC", "CCA", "CCG"], "S": ["TCT", "TCC", "TCA", "TCG", "AGT", "AGC"], "T": ["ACT", "ACC", "ACA", "ACG"], "W": ["TGG"],
    "Y": ["TAT", "TAC"], "V": ["GTT", "GTC", "GTA", "GTG"], "X": ["NNN"],
}
AA = "ARNDCQEGHILKMFPSTWYV"


def splitmix64(x):
    """Vectorised splitmix64 finaliser on uint64 arrays."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


class Counter:
    """Stream of uint64 values: value k = splitmix64(seed * K + k)."""

    def __init__(self, seed):
        self.base = np.uint64(splitmix64(np.array([seed], dtype=np.uint64))[0])
        self.k = 0

    def take(self, n):
        with np.errstate(over="ignore"):
            idx = (np.arange(self.k, self.k + n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + self.base) & _M64
        self.k += n
        return splitmix64(idx)

    def uniform(self, n):
        return (self.take(n) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)

    def integers(self, n, hi):
        return (self.take(n) % np.uint64(hi)).astype(np.int64)


def build_genomes(marker_seqs, total_bp=8_000_000, seed=20261001, marker_gene_fraction=0.01, divergence=0.25):
    """Returns one uint8 array of bases (all genomes concatenated)."""
    rng = Counter(seed)
    comp = np.zeros(20)
    sample = "".join(marker_seqs[:: max(1, len(marker_seqs) // 500)])
    for i, a in enumerate(AA):
        comp[i] = sample.count(a)
    cdf = np.cumsum(comp / comp.sum())
    codon_tab = {a: [np.frombuffer(c.encode(), dtype=np.uint8) for c in cs] for a, cs in CODONS.items()}
    rc_map = np.zeros(256, dtype=np.uint8)
    for a, b in zip(b"ACGTN", b"TGCAN"):
        rc_map[a] = b
    parts, n = [], 0
    while n < total_bp:
        u = rng.uniform(4)
        if u[0] < marker_gene_fraction:
            prot = marker_seqs[int(u[1] * len(marker_seqs))].replace("X", "A")
            sub = rng.uniform(len(prot))
            repl = np.searchsorted(cdf, rng.uniform(len(prot)))
            prot = "".join(AA[min(19, int(repl[i]))] if sub[i] < divergence else c for i, c in enumerate(prot))
        else:
            ln = 100 + int(u[1] * 500)
            prot = "".join(AA[min(19, int(k))] for k in np.searchsorted(cdf, rng.uniform(ln)))
        pick = rng.take(len(prot))
        gene = np.concatenate([codon_tab[a][int(pick[i] % np.uint64(len(codon_tab[a])))] for i, a in enumerate(prot)])
        if u[2] < 0.5:
            gene = rc_map[gene[::-1]]
        spacer = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(20 + int(u[3] * 120), 4)]
        parts += [gene, spacer]
        n += len(gene) + len(spacer)
    return np.concatenate(parts)


def sample_reads(genome, nreads, read_len, seed=1, chunk=1 << 20):
    """(nreads, read_len) uint8 bases, uniform start, strand by fair coin, error free."""
    rng = Counter(seed)
    out = np.empty((nreads, read_len), dtype=np.uint8)
    rc_map = np.zeros(256, dtype=np.uint8)
    for a, b in zip(b"ACGTN", b"TGCAN"):
        rc_map[a] = b
    span = len(genome) - read_len
    ar = np.arange(read_len, dtype=np.int64)
    for s in range(0, nreads, chunk):
        m = min(chunk, nreads - s)
        start = rng.integers(m, span)
        strand = rng.take(m) & np.uint64(1)
        block = genome[start[:, None] + ar[None, :]]
        rev = strand.astype(bool)
        block[rev] = rc_map[block[rev][:, ::-1]]
        out[s:s + m] = block
    return out


def random_proteins(total_residues, seed=20261003, marker_seqs=None, lo=100, hi=600):
    """A protein database of random ORFs with the marker proteins' amino-acid composition: (names, seqs).  6.5 M residues fill more
    than half of the 10^6 reduced-alphabet 6-mer buckets, so the database's `.info` threshold (median bucket size) is above 0 and
    RAPsearch2 lengthens the seeds of frequent buckets by letter frequency (`Searching 0x415ec0-0x415f71`): the generic seed
    path, which the marker database (threshold 0) never takes."""
    rng = Counter(seed)
    comp = np.zeros(20)
    if marker_seqs is None:
        comp[:] = [8.6, 5.9, 3.9, 5.4, 0.9, 3.6, 6.9, 7.6, 2.1, 6.6, 9.3, 6.3, 2.5, 3.7, 4.3, 5.6, 5.4, 1.0, 2.9, 7.5]
    else:
        sample = "".join(marker_seqs[:: max(1, len(marker_seqs) // 500)])
        for i, a in enumerate(AA):
            comp[i] = sample.count(a)
    cdf = np.cumsum(comp / comp.sum())
    aa = np.frombuffer(AA.encode(), dtype=np.uint8)
    names, seqs, n = [], [], 0
    while n < total_residues:
        ln = lo + int(rng.integers(1, hi - lo)[0])
        seqs.append(aa[np.minimum(19, np.searchsorted(cdf, rng.uniform(ln)))].tobytes().decode())
        names.append("orf%06d" % len(names))
        n += ln
    return names, seqs


def mutate_reads(reads, read_len, sub_rate=0.02, indel_rate=0.005, seed=1):
    """Sequencing-error model for parity tests: every base of `reads` (n x >= read_len + slack, error free) is substituted by
    one of the other three with probability sub_rate, deleted with probability indel_rate / 2, or followed by a random inserted
    base with probability indel_rate / 2; the result is trimmed to read_len (reads that got too short are padded with `N`).
    An indel inside a codon shifts the frame: the gapped extension has to bridge between frames' HSPs or stop there."""
    rng = Counter(seed)
    n, w = reads.shape
    u = rng.uniform(n * w).reshape(n, w)
    pick = rng.integers(n * w, 3).reshape(n, w)
    ins = rng.integers(n * w, 4).reshape(n, w)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    code = np.zeros(256, dtype=np.int64)
    for i, c in enumerate(b"ACGT"):
        code[c] = i
    out = np.full((n, read_len), ord("N"), dtype=np.uint8)
    sub = u < sub_rate
    dele = (u >= sub_rate) & (u < sub_rate + indel_rate / 2)
    insr = (u >= sub_rate + indel_rate / 2) & (u < sub_rate + indel_rate)
    subbed = np.where(sub, acgt[(code[reads] + 1 + pick) & 3], reads)
    for i in range(n):
        if not (dele[i].any() or insr[i].any()):
            row = subbed[i]
        else:
            keep = ~dele[i]
            reps = keep.astype(np.int64) + insr[i]
            row = np.repeat(subbed[i], reps)
            at = np.cumsum(reps)[insr[i]] - 1              # the copy behind an insertion point becomes the inserted base
            row[at] = acgt[ins[i][insr[i]]]
        m = min(read_len, len(row))
        out[i, :m] = row[:m]
    return out


# ---------------------------------------------------------------------------------------------------------------------
# Reads from the reference's 30 real genomes; data fixture written by tests/golden/make_genomes_fixture.py
# ---------------------------------------------------------------------------------------------------------------------
GENOMES_NPZ = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "genomes", "genomes30.npz")


def load_genomes(path=None):
    """(bases uint8 [total], contig_off int64 [ncontig + 1]) of the 30 genomes (251 contigs, 84.8 Mbp)."""
    z = np.load(path or GENOMES_NPZ)
    p = z["packed"]
    b = np.empty((len(p), 4), dtype=np.uint8)
    for k in range(4):
        b[:, k] = (p >> (2 * k)) & 3
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)[b.reshape(-1)[: int(z["contig_off"][-1])]]
    bases[z["exc_pos"]] = z["exc_chr"]
    return bases, z["contig_off"].astype(np.int64)


def _t_splitmix64(x):
    import torch
    M = lambda v: torch.tensor(v, dtype=torch.int64, device=x.device)  # noqa: E731
    lsr = lambda v, k: (v >> k) & ((1 << (64 - k)) - 1)                # noqa: E731  logical shift on int64
    x = x + M(-7046029254386353131)                  # 0x9E3779B97F4A7C15
    z = (x ^ lsr(x, 30)) * M(-4658895280553007687)   # 0xBF58476D1CE4E5B9
    z = (z ^ lsr(z, 27)) * M(-7723592293110705685)   # 0x94D049BB133111EB
    return z ^ lsr(z, 31)


class GenomeReads:
    """Error-free shotgun reads of the 30 genomes, generated with torch on any device (a pure function of (seed, read
    index), so every rank / device / chunking produces the same read for the same index).  Start positions are uniform
    over the positions of every contig that can hold the whole read (or fragment); strand by fair coin.

    single(n, L): n reads.   paired(n, L, frag): n fragments of `frag` bases -> (mate1, mate2), mate 1 = the first L bases
    of the fragment, mate 2 = the reverse complement of its last L bases (BASELINE configs[3]; the reference consumes the two
    files one after the other, /root/reference/microbe_census/microbe_census.py:337-356)."""

    def __init__(self, device="cpu", seed=20261001, path=None):
        import torch
        bases, off = load_genomes(path)
        self.torch, self.device, self.seed = torch, torch.device(device), int(seed)
        self.g = torch.from_numpy(bases).to(self.device)
        self.off = torch.from_numpy(off).to(self.device)
        self.clen = self.off[1:] - self.off[:-1]
        rc = torch.zeros(256, dtype=torch.uint8)
        for a, b in zip(b"ACGTN", b"TGCAN"):
            rc[a] = b
        for c in range(256):
            if rc[c] == 0:
                rc[c] = ord("N")
        self.rc = rc.to(self.device)

    def _starts(self, first, n, span):
        torch = self.torch
        valid = torch.clamp(self.clen - span + 1, min=0)
        cum = torch.cumsum(valid, 0)
        total = int(cum[-1].item())
        idx = torch.arange(first, first + n, device=self.device, dtype=torch.int64)
        v = _t_splitmix64(_t_splitmix64(idx + self.seed * 1000003))
        u = ((v >> 1) & 0x7FFFFFFFFFFFFFFF) % total
        c = torch.searchsorted(cum, u, right=True)
        start = self.off[c] + (u - (cum[c] - valid[c]))
        return start, (v & 1).bool()

    def _take(self, start, rev, span, L):
        """first L bases of the (strand-oriented) fragment [start, start+span)"""
        torch = self.torch
        ar = torch.arange(L, device=self.device, dtype=torch.int64)
        fwd = self.g[start[:, None] + ar[None, :]]
        back = self.rc[self.g[(start + span - 1)[:, None] - ar[None, :]].long()]
        return torch.where(rev[:, None], back, fwd)

    def single(self, n, L, first=0, chunk=1 << 20):
        out = self.torch.empty((n, L), dtype=self.torch.uint8, device=self.device)
        for s in range(0, n, chunk):
            m = min(chunk, n - s)
            start, rev = self._starts(first + s, m, L)
            out[s:s + m] = self._take(start, rev, L, L)
        return out

    def paired(self, n, L, frag=400, first=0, chunk=1 << 20):
        torch = self.torch
        m1 = torch.empty((n, L), dtype=torch.uint8, device=self.device)
        m2 = torch.empty((n, L), dtype=torch.uint8, device=self.device)
        for s in range(0, n, chunk):
            m = min(chunk, n - s)
            start, rev = self._starts(first + s, m, frag)
            m1[s:s + m] = self._take(start, rev, frag, L)
            m2[s:s + m] = self._take(start, ~rev, frag, L)          # the other end, on the other strand
        return m1, m2
