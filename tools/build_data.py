#!/usr/bin/env python3
"""Build the package data of microbecensus_amd from the reference's DATA files (build container only).

Inputs (data, not code) under /root/reference:
  training/input/gene_fams/*.faa.gz        the marker proteins - the only surviving source of the
                                           missing `microbe_census/data/rapdb_2.15`
  microbe_census/data/{gene_fam,gene_len,pars,coefficients,weights,read_len}.map

Outputs:
  microbecensus_amd/data/markers.faa.gz    canonical marker FASTA: files in sorted() order, records in
                                           file order, first occurrence of each distinct sequence
  microbecensus_amd/data/model.json        family of every marker (aligned with the FASTA order), the
                                           per-(family, read length) classification parameters, and the
                                           AGS model coefficients / weights
The lookups keep the reference's dictionary semantics (read_dic, microbe_census.py:74-88: the LAST
line of a duplicated key wins).
"""
import gzip
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(REPO, "oracle"))
from build_ref import REF, canonical_markers  # noqa: E402

OUT = os.path.join(REPO, "microbecensus_amd", "data")


def read_map(name):
    d = {}
    with open(os.path.join(REF, "microbe_census/data", name)) as f:
        for line in f:
            k, v = line.rstrip().split()
            d[k] = v
    return d


def main():
    os.makedirs(OUT, exist_ok=True)
    markers = list(canonical_markers())
    with gzip.GzipFile(os.path.join(OUT, "markers.faa.gz"), "wb", mtime=0) as f:
        for name, seq in markers:
            f.write((">%s\n%s\n" % (name, seq)).encode())
    gene_fam = read_map("gene_fam.map")
    gene_len = read_map("gene_len.map")
    families = sorted(set(gene_fam.values()))
    for name, seq in markers:
        assert float(gene_len[name]) == len(seq), name
    pars = {}
    with open(os.path.join(REF, "microbe_census/data/pars.map")) as f:
        next(f)
        for line in f:
            fam, L, cov, aaid, score, stat = line.rstrip().split()
            pars.setdefault(L, {})[fam] = [float(cov), float(aaid), float(score), stat]
    model = {
        "families": families,
        "marker_family": [families.index(gene_fam[n]) for n, _ in markers],
        "read_lengths": [int(x) for x in open(os.path.join(REF, "microbe_census/data/read_len.map")).read().split()],
        "pars": pars,
        "coefficients": {k: float(v) for k, v in read_map("coefficients.map").items()},
        "weights": {k: float(v) for k, v in read_map("weights.map").items()},
    }
    with open(os.path.join(OUT, "model.json"), "w") as f:
        json.dump(model, f, separators=(",", ":"), sort_keys=True)
    print("markers", len(markers), "families", len(families), "lengths", len(model["read_lengths"]))


if __name__ == "__main__":
    main()
