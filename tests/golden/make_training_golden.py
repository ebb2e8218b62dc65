#!/usr/bin/env python3
"""Golden vector for the training workflow's grid classification, produced by RUNNING THE REFERENCE's own
functions here: training/training.py:210-219 (parse_rapsearch), :229-334 (read_hits, aln_filter, pid_filter, score_filter,
find_best_hits, aggregate_hits, classify_reads) and :336-343 (drange) are executed UNCHANGED - their source text is read from
/root/reference at generation time and exec'd (the module itself cannot be imported: it needs Biopython) - on the reference's
own m8 of the unit-test metagenome (tests/golden/unittest_metagenome.m8.gz), with the grid of training/class_reads.py:51-53 and
the reference's gene_fam.map / gene_len.map.  The only Python-2-ism inside those lines is `.iteritems()` on the dict
aggregate_hits returns (:331): the generator hands classify_reads an aggregate_hits that returns a dict subclass with that
method - no line of the reference is edited.

Output: tests/golden/training_grid_unittest.json.gz - the rows of the .hits table with count_hits > 0 as
[fam, aln_cov, max_pid, min_score, count_hits, count_aln, count_cov].  Only runs where /root/reference exists."""
import gzip
import json
import os
import tempfile

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


class Py2Dict(dict):
    def iteritems(self):
        return iter(self.items())


def main():
    src = open(os.path.join(REF, "training", "training.py")).read().split("\n")
    # (1-based, inclusive) parse_rapsearch 210-219, read_hits .. classify_reads 229-334, drange 336-343
    text = "\n".join(src[209:219] + [""] + src[228:334] + [""] + src[335:343]) + "\n"
    ns = {}
    exec(compile(text, "training.py[210-219,229-334,336-343]", "exec"), ns)
    ref_aggregate = ns["aggregate_hits"]
    ns["aggregate_hits"] = lambda *a, **k: Py2Dict(ref_aggregate(*a, **k))
    data = os.path.join(REF, "microbe_census", "data")
    gene2fam = dict(line.split() for line in open(os.path.join(data, "gene_fam.map")))
    gene2len = {k: int(v) for k, v in (line.split() for line in open(os.path.join(data, "gene_len.map")))}
    fams = set(gene2fam.values())
    aln_covs, max_pids, min_scores = [0.00, 0.25, 0.50, 0.75], [50, 60, 70, 80, 90, 100], ns["drange"](23, 50, 1)
    with tempfile.TemporaryDirectory() as td:
        m8 = os.path.join(td, "unittest.m8")
        with open(m8, "wb") as f:
            f.write(gzip.open(os.path.join(HERE, "unittest_metagenome.m8.gz"), "rb").read())
        out = os.path.join(td, "unittest.hits")
        ns["classify_reads"](m8, out, aln_covs, max_pids, min_scores, gene2len, gene2fam, fams, "100")   # class_reads.py passes the directory name
        rows = []
        with open(out) as f:
            head = f.readline().split()
            assert head == ["fam", "aln_cov", "max_pid", "min_score", "count_hits", "count_aln", "count_cov"]
            for line in f:
                x = line.split()
                if int(x[4]) > 0:
                    rows.append([x[0], float(x[1]), int(x[2]), float(x[3]), int(x[4]), int(x[5]), float(x[6])])
    rows.sort()
    doc = {"source": "training/training.py:311-334 classify_reads on tests/golden/unittest_metagenome.m8.gz, read_length '100'",
           "aln_covs": aln_covs, "max_pids": max_pids, "min_scores": [float(v) for v in min_scores], "n_rows_with_hits": len(rows), "rows": rows}
    with gzip.open(os.path.join(HERE, "training_grid_unittest.json.gz"), "wt") as f:
        json.dump(doc, f)
    print("wrote %d rows with hits; total hits %d" % (len(rows), sum(r[4] for r in rows)))


if __name__ == "__main__":
    main()
