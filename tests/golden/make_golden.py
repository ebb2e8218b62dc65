#!/usr/bin/env python3
"""Generate the committed golden vectors by RUNNING THE REFERENCE in this container.

Only runs where /root/reference exists (never on the GPU box).  It imports the reference's
Python module from a scratch copy (the tree is read-only and lacks the DB blob), drops the
DB rebuilt by oracle/build_ref.py next to it and calls the reference's own stage functions
(process_seqfile -> search_seqs -> classify_reads -> aggregate_hits ->
estimate_average_genome_size, microbe_census.py:586-631), capturing what each stage produced.

Outputs (data only - inputs and expected outputs):
  tests/golden/<case>.json      args, sampled_reads, best_hits, agg_hits, est_ags, md5s
  tests/golden/<case>.m8.gz     the RAPsearch2 m8 (non-# lines)
  tests/golden/<case>.reads.fa.gz   the trimmed reads the reference fed to the search
  tests/golden/inputs/*         the reference's own test/example input files
"""
import gzip
import hashlib
import importlib
import json
import os
import shutil
import sys
import tempfile

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
ORACLE_REF = os.path.join(REPO, "oracle", "_ref")


def load_reference():
    scratch = tempfile.mkdtemp(prefix="mc_ref_")
    shutil.copytree(os.path.join(REF, "microbe_census"), os.path.join(scratch, "microbe_census"))
    data = os.path.join(scratch, "microbe_census", "data")
    os.chmod(data, 0o755)
    for f in ("rapdb_2.15", "rapdb_2.15.info"):
        dst = os.path.join(data, f)
        if os.path.exists(dst):
            os.chmod(dst, 0o644)
            os.remove(dst)
        os.symlink(os.path.join(ORACLE_REF, f), dst)
    for b in os.listdir(os.path.join(scratch, "microbe_census", "bin")):
        os.chmod(os.path.join(scratch, "microbe_census", "bin", b), 0o755)
    sys.path.insert(0, scratch)
    mod = importlib.import_module("microbe_census.microbe_census")
    return mod, scratch


def md5_bytes(b):
    return hashlib.md5(b).hexdigest()


def run_case(mc, name, args):
    paths = mc.get_relative_paths(args)
    mc.check_paths(paths)
    mc.check_input(args)
    mc.impute_missing_args(args)
    mc.check_arguments(args)
    mc.process_seqfile(args, paths)
    reads = open(paths["tempfile"], "rb").read()
    mc.search_seqs(args, paths)
    m8 = b"".join(l for l in open(paths["tempfile"] + ".m8", "rb") if not l.startswith(b"#"))
    best_hits = mc.classify_reads(args, paths)
    agg_hits = mc.aggregate_hits(args, paths, best_hits)
    mc.clean_up(paths)
    est = mc.estimate_average_genome_size(args, paths, agg_hits)
    out = {
        "case": name,
        "args": {k: v for k, v in args.items() if k != "seqfiles"},
        "seqfiles": [os.path.basename(p) for p in args["seqfiles"]],
        "sampled_reads": args["sampled_reads"],
        "m8_rows": m8.count(b"\n"),
        "m8_md5": md5_bytes(m8),
        "reads_md5": md5_bytes(reads),
        "best_hits": {k: v for k, v in best_hits.items()},
        "agg_hits": agg_hits,
        "est_ags": est,
    }
    with open(os.path.join(HERE, name + ".json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    with gzip.GzipFile(os.path.join(HERE, name + ".m8.gz"), "wb", mtime=0) as f:
        f.write(m8)
    with gzip.GzipFile(os.path.join(HERE, name + ".reads.fa.gz"), "wb", mtime=0) as f:
        f.write(reads)
    print(name, "sampled", args["sampled_reads"], "m8 rows", out["m8_rows"], out["m8_md5"],
          "best_hits", len(best_hits), "AGS", repr(est))
    return out


def write_synthetic_inputs(inputs):
    """Small versions of BASELINE configs[1], [3] and [4] from the repo's own read generator over the 30-genome fixture
    (microbecensus_amd/synth.py GenomeReads; data only): 100 bp FASTA; a paired 150 bp FASTQ library in two files; 300 bp FASTQ
    with SURVEY 8(d)'s quality model, 5 % of the reads with one base below 20, 2 % exact and 1 % reverse-complement duplicates."""
    import numpy as np
    sys.path.insert(0, REPO)
    from microbecensus_amd import synth
    gen = synth.GenomeReads(device="cpu", seed=20261001)

    def gz_write(name, data):
        with gzip.GzipFile(os.path.join(inputs, name), "wb", mtime=0) as f:
            f.write(data)
    r = gen.single(30000, 100, first=10_000_000).numpy()
    gz_write("c2_100bp.fa.gz", b"".join(b">r%d\n%s\n" % (i, bytes(x)) for i, x in enumerate(r)))
    m1, m2 = gen.paired(12000, 150, frag=300, first=20_000_000)
    q = b"I" * 150
    gz_write("c4_pair_1.fq.gz", b"".join(b"@f%d/1\n%s\n+\n%s\n" % (i, bytes(x), q) for i, x in enumerate(m1.numpy())))
    gz_write("c4_pair_2.fq.gz", b"".join(b"@f%d/2\n%s\n+\n%s\n" % (i, bytes(x), q) for i, x in enumerate(m2.numpy())))
    r = gen.single(6000, 300, first=30_000_000).numpy()
    rng = np.random.RandomState(20261001)
    comp = bytes.maketrans(b"ACGTN", b"TGCAN")
    recs, pool = [], []
    for i, x in enumerate(r):
        sq = bytes(x)
        u = rng.rand()
        if pool and u < 0.02:
            sq = pool[rng.randint(len(pool))]
        elif pool and u < 0.03:
            sq = pool[rng.randint(len(pool))][::-1].translate(comp)
        else:
            pool.append(sq)
        ql = np.clip(np.rint(rng.normal(34, 6, size=300)), 20, 41).astype(np.int64)
        if rng.rand() < 0.05:
            ql[rng.randint(300)] = rng.choice([2, 10, 19])
        recs.append(b"@s%d\n%s\n+\n%s\n" % (i, sq, bytes((ql + 33).astype(np.uint8))))
    gz_write("c5_300bp.fq.gz", b"".join(recs))


def main():
    mc, scratch = load_reference()
    inputs = os.path.join(HERE, "inputs")
    os.makedirs(inputs, exist_ok=True)
    if not os.path.exists(os.path.join(inputs, "c5_300bp.fq.gz")):
        write_synthetic_inputs(inputs)
    for src in ("tests/data/metagenome.fa.gz", "microbe_census/example/example.fq.gz",
                "microbe_census/example/example.fa.gz"):
        dst = os.path.join(inputs, os.path.basename(src))
        if not os.path.exists(dst):
            shutil.copyfile(os.path.join(REF, src), dst)
            os.chmod(dst, 0o644)
    # reference unit test (tests/test_microbe_census.py:15-25): all defaults
    run_case(mc, "unittest_metagenome", {"seqfiles": [os.path.join(inputs, "metagenome.fa.gz")]})
    # BASELINE config 1: example.fq.gz -n 10000 -l 100 -t 1
    a = {"seqfiles": [os.path.join(inputs, "example.fq.gz")], "nreads": 10000, "read_length": 100,
         "threads": 1}
    out = run_case(mc, "config1_example_fq", a)
    a["seqfiles"] = [os.path.join(inputs, "example.fq.gz")]
    tb = mc.count_bases(a)
    out["total_bases"] = tb
    out["genome_equivalents"] = tb / out["est_ags"]
    with open(os.path.join(HERE, "config1_example_fq.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    # small versions of BASELINE configs[1], [3], [4] (synthetic inputs written above)
    run_case(mc, "c2_100bp", {"seqfiles": [os.path.join(inputs, "c2_100bp.fa.gz")], "threads": 8})
    run_case(mc, "c4_paired", {"seqfiles": [os.path.join(inputs, "c4_pair_1.fq.gz"), os.path.join(inputs, "c4_pair_2.fq.gz")], "threads": 8, "nreads": 20000})
    run_case(mc, "c5_300bp_q20_dups", {"seqfiles": [os.path.join(inputs, "c5_300bp.fq.gz")], "threads": 8, "min_quality": 20, "filter_dups": True})
    # extra cases supplied on the command line:  name=path[,path]:key=val:key=val
    for spec in sys.argv[1:]:
        name, rest = spec.split("=", 1)
        parts = rest.split(":")
        args = {"seqfiles": [os.path.abspath(p) for p in parts[0].split(",")]}
        for kv in parts[1:]:
            k, v = kv.split("=")
            args[k] = (v == "True") if v in ("True", "False") else int(v)
        run_case(mc, name, args)
    shutil.rmtree(scratch, ignore_errors=True)


if __name__ == "__main__":
    main()
