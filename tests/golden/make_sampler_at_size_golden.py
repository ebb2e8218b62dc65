#!/usr/bin/env python3
"""Golden vector of the read sampler AT SIZE, produced by RUNNING THE REFERENCE's own process_seqfile
(microbe_census.py:328-367, with quality_filter :265-279 and parse_seqs :294-325) here on the 150,000-record FASTQ file of
tests/golden/c5_at_size.py with `-q 20 -d` (BASELINE configs[4] shape): the four counters, args['sampled_reads'], the md5 of the
temp FASTA it wrote, count_bases().  Output: tests/golden/c5_at_size.json.  Only runs where /root/reference exists."""
import contextlib
import hashlib
import io
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from make_golden import load_reference   # noqa: E402
import c5_at_size                        # noqa: E402


def main():
    mc, scratch = load_reference()
    with tempfile.TemporaryDirectory() as td:
        fq = os.path.join(td, "c5.fq")
        n = c5_at_size.write_fastq(fq)
        args = {"seqfiles": [fq], "min_quality": 20, "filter_dups": True, "nreads": 10_000_000, "verbose": True}
        paths = mc.get_relative_paths(args)
        mc.check_input(args)
        mc.impute_missing_args(args)
        mc.check_arguments(args)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            mc.process_seqfile(args, paths)
        lines = [l.strip() for l in buf.getvalue().split("\n") if l.startswith("\t")]
        counters = {"too_short": int(lines[0].split()[0]), "low_qual": int(lines[1].split()[0]), "dups": int(lines[2].split()[0]), "sampled": int(lines[3].split()[0])}
        fasta = open(paths["tempfile"], "rb").read()
        doc = {"source": "microbe_census.py:328-367 process_seqfile on tests/golden/c5_at_size.py write_fastq(), -q 20 -d", "records": n,
               "file_md5": hashlib.md5(open(fq, "rb").read()).hexdigest(), "read_length": args["read_length"], "quality_offset": args["quality_offset"],
               "sampled_reads": args["sampled_reads"], "counters": counters, "reads_md5": hashlib.md5(fasta).hexdigest(), "count_bases": mc.count_bases(dict(args, verbose=False))}
        mc.clean_up(paths)
    assert counters["sampled"] == doc["sampled_reads"]
    json.dump(doc, open(os.path.join(HERE, "c5_at_size.json"), "w"), indent=1, sort_keys=True)
    print(doc)


if __name__ == "__main__":
    main()
