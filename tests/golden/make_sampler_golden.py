#!/usr/bin/env python3
"""Golden vectors for the read sampler (process_seqfile / count_bases), produced by RUNNING THE REFERENCE's Python here.

Writes small synthetic input files (this script is their only source; they exercise the parser's quirks: CRLF and lone-CR
line ends, missing final newline, multi-line records, '>' / '@' inside qualities, blank lines, lowercase and IUPAC bases,
N runs, short reads, exact and reverse-complement duplicates, low qualities, truncated qualities) into
tests/golden/sampler/ and records, per (files, arguments) case, what the reference produced: the four counters it prints,
the temp FASTA it wrote, count_bases() - or the exception class it raised.  Only runs where /root/reference exists.
"""
import contextlib
import gzip
import io
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import load_reference   # noqa: E402

OUT = os.path.join(HERE, "sampler")


def rc(s):
    return s[::-1].translate(str.maketrans("ACGTN", "TGCAN"))


def write_inputs():
    rnd = random.Random(20261001)
    dna = lambda n: "".join(rnd.choice("ACGT") for _ in range(n))   # noqa: E731
    files = {}
    # 1. FASTA, multi-line, mixed lengths, N runs, duplicates (exact + reverse complement), a lowercase read
    recs = []
    base = [dna(rnd.choice([40, 60, 75, 80, 120, 151])) for _ in range(60)]
    for i, s in enumerate(base):
        recs.append(("r%d some description\twith tab" % i, s))
    recs.insert(10, ("dup_exact", base[3]))
    recs.insert(20, ("dup_rc", rc(base[5])))
    recs.insert(25, ("many_n", "N" * 30 + dna(70)))
    recs.insert(26, ("few_n", "N" * 2 + dna(98)))
    recs.insert(30, ("lower", dna(90).lower()))
    recs.insert(31, ("empty", ""))
    txt = ""
    for name, s in recs:
        txt += ">%s\n" % name
        for k in range(0, len(s), 37):
            txt += s[k:k + 37] + "\n"
    files["a.fa"] = txt.encode()
    files["a_crlf.fa"] = txt.replace("\n", "\r\n").encode()
    files["a_cr.fa"] = txt.replace("\n", "\r").encode()
    files["a_nofinalnl.fa"] = txt[:-1].encode()
    files["a_blank.fa"] = txt.replace(">r7 ", "\n>r7 ").replace(">r9 ", "\n\n>r9 ").encode()
    # 2. FASTQ, single and multi-line, '@' and '>' as quality characters, low qualities, a truncated last record
    q_ok = lambda n: "".join(rnd.choice("FGHIJ") for _ in range(n))   # noqa: E731
    fq = ""
    for i in range(50):
        n = rnd.choice([50, 77, 100, 101, 150])
        s = dna(n)
        q = q_ok(n)
        if i % 7 == 0:
            q = q[:5] + "#" + q[6:]                      # one very low quality inside the first bases
        if i % 11 == 0:
            q = "@" + q[1:]                              # quality line starting with '@'
        if i % 13 == 0:
            q = ">" + q[1:]
        if i % 9 == 0:                                   # multi-line sequence and quality
            fq += "@q%d extra\n%s\n%s\n+\n%s\n%s\n" % (i, s[:33], s[33:], q[:20], q[20:])
        else:
            fq += "@q%d\n%s\n+q%d\n%s\n" % (i, s, i, q)
    files["b.fq"] = fq.encode()
    files["b.fq.gz"] = gzip.compress(fq.encode())
    files["b_trunc.fq"] = (fq + "@last\n" + dna(80) + "\n+\n" + q_ok(40)).encode()
    # a lone '+', '>' or '@' WITHOUT newline as the last line: `last = l[:-1]` leaves '' and the parser stops after the current record
    files["b_loneplus.fq"] = (fq + "@last\n" + dna(80) + "\n+").encode()
    files["a_lonegt.fa"] = (txt + ">").encode()
    files["a_loneat.fa"] = (txt + "@").encode()
    files["a_onlygt.fa"] = (">r0\n" + dna(60) + "\n\n@").encode()
    files["b_iupac.fa"] = (">x1\n" + dna(60) + "\n>x2\n" + dna(30) + "RYK" + dna(40) + "\n>x3\n" + dna(70) + "\n").encode()
    # 4. BASELINE configs[4] in small: 300+ bp FASTQ, phred+33 qualities ~ clipped N(34, 6), 5 % of the reads with one base < 20 inside
    #    the first 300, 2 % exact duplicates, 1 % reverse-complement duplicates, a few short reads
    d, pool = "", []
    for i in range(600):
        u = rnd.random()
        if pool and u < 0.02:
            s = rnd.choice(pool)
        elif pool and u < 0.03:
            s = rc(rnd.choice(pool))
        else:
            s = dna(rnd.choice([250, 300, 301, 320, 350]))
            pool.append(s)
        q = [min(41, max(20, int(round(rnd.gauss(34, 6))))) for _ in s]
        if rnd.random() < 0.05:
            q[rnd.randrange(min(len(s), 300))] = rnd.choice([2, 10, 19])
        d += "@d%d\n%s\n+\n%s\n" % (i, s, "".join(chr(33 + v) for v in q))
    files["d.fq.gz"] = gzip.compress(d.encode())
    # 3. second FASTA for multi-file sampling
    files["c.fa.gz"] = gzip.compress("".join(">c%d\n%s\n" % (i, dna(100)) for i in range(40)).encode())
    os.makedirs(OUT, exist_ok=True)
    for name, data in files.items():
        with open(os.path.join(OUT, name), "wb") as f:
            f.write(data)
    return sorted(files)


CASES = [
    # name, files, args
    ("fa_default", ["a.fa"], {"read_length": 50}),
    ("fa_len75", ["a.fa"], {"read_length": 75}),
    ("fa_n10", ["a.fa"], {"read_length": 50, "nreads": 10}),
    ("fa_dups", ["a.fa"], {"read_length": 50, "filter_dups": True}),              # lowercase read + -d: KeyError
    ("fa_unknown5", ["a.fa"], {"read_length": 100, "max_unknown": 5}),
    ("fa_unknown1", ["a.fa"], {"read_length": 100, "max_unknown": 1}),
    ("fa_crlf", ["a_crlf.fa"], {"read_length": 50}),
    ("fa_cr", ["a_cr.fa"], {"read_length": 50}),
    ("fa_nofinalnl", ["a_nofinalnl.fa"], {"read_length": 50}),
    ("fa_blank", ["a_blank.fa"], {"read_length": 50}),
    ("fq_default", ["b.fq"], {"read_length": 50}),
    ("fq_gz", ["b.fq.gz"], {"read_length": 75}),
    ("fq_minq20", ["b.fq"], {"read_length": 50, "min_quality": 20}),
    ("fq_meanq39", ["b.fq"], {"read_length": 50, "mean_quality": 39}),
    ("fq_meanq40", ["b.fq"], {"read_length": 100, "mean_quality": 40, "min_quality": 3}),
    ("fq_trunc", ["b_trunc.fq"], {"read_length": 50}),
    ("fq_dups", ["b.fq", "b.fq.gz"], {"read_length": 50, "filter_dups": True}),
    ("multi", ["c.fa.gz", "a.fa"], {"read_length": 50, "nreads": 55}),
    ("multi_dups", ["c.fa.gz", "c.fa.gz"], {"read_length": 100, "filter_dups": True}),
    ("iupac_dups", ["b_iupac.fa"], {"read_length": 50, "filter_dups": True}),
    ("none_left", ["a.fa"], {"read_length": 500}),
    ("fq_lone_plus", ["b_loneplus.fq"], {"read_length": 50}),                      # last record has no qualities: TypeError in quality_filter
    ("fq_lone_plus_n20", ["b_loneplus.fq"], {"read_length": 50, "nreads": 20}),
    ("fa_lone_gt", ["a_lonegt.fa"], {"read_length": 50}),
    ("fa_lone_at", ["a_loneat.fa", "c.fa.gz"], {"read_length": 50}),
    ("fa_only_gt", ["a_onlygt.fa"], {"read_length": 50}),
    ("fa_nreads_none", ["a.fa", "c.fa.gz"], {"read_length": 50, "nreads": None}),  # None = no cap (read_id == None is never true)
    ("fq_q20_dups_300", ["d.fq.gz"], {"read_length": 300, "min_quality": 20, "filter_dups": True}),   # BASELINE configs[4] in small
]


def main():
    mc, scratch = load_reference()
    names = write_inputs()
    out = {"inputs": names, "cases": []}
    for name, files, extra in CASES:
        args = {"seqfiles": [os.path.join(OUT, f) for f in files], "verbose": True}
        args.update(extra)
        res = {"case": name, "files": files, "args": dict(extra)}
        buf = io.StringIO()
        try:
            with contextlib.redirect_stdout(buf):
                paths = mc.get_relative_paths(args)
                mc.check_input(args)
                mc.impute_missing_args(args)
                mc.check_arguments(args)
                try:
                    mc.process_seqfile(args, paths)
                    res["sampled_reads"] = args["sampled_reads"]
                    res["temp_fasta"] = open(paths["tempfile"]).read()
                finally:
                    mc.clean_up(paths)
            res["file_type"], res["quality_offset"] = args["file_type"], args.get("quality_offset")
            lines = [l.strip() for l in buf.getvalue().splitlines()]
            res["too_short"] = int([l for l in lines if "shorter than" in l][0].split()[0])
            res["low_qual"] = int([l for l in lines if "low quality" in l][0].split()[0])
            res["dups"] = int([l for l in lines if "duplicate reads" in l][0].split()[0])
        except SystemExit as e:
            res["exit"] = str(e)
        except Exception as e:   # run_pipeline would print and swallow this
            res["raises"] = type(e).__name__
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                res["count_bases"] = mc.count_bases({"seqfiles": args["seqfiles"], "verbose": False})
        except Exception as e:
            res["count_bases_raises"] = type(e).__name__
        out["cases"].append(res)
        print(name, {k: v for k, v in res.items() if k not in ("temp_fasta", "args", "files")})
    with open(os.path.join(HERE, "sampler_cases.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
