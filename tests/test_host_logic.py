"""Host-side mirror of the reference's Python interface (microbecensus_amd/microbe_census.py) against the
golden vectors captured from the reference, plus the reference's own unit tests restated
(tests/test_microbe_census.py:27-82 of the reference: ReadList, FileType)."""
import gzip
import hashlib
import json
import os

import pytest

from microbecensus_amd import microbe_census as mc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
INPUTS = os.path.join(GOLD, "inputs")


def golden(case):
    return json.load(open(os.path.join(GOLD, case + ".json")))


def test_read_list(tmp_path):
    p = tmp_path / "tmp.txt"
    p.write_text("".join("%d\n" % i for i in range(10)))
    assert mc.read_list(str(p), header=False, dtype="int") == list(range(10))


def test_detect_filetype(tmp_path):
    fq = tmp_path / "a.fastq"; fq.write_text("@r1\nACGT\n+\nIIII\n")
    fa = tmp_path / "a.fasta"; fa.write_text(">r1\nACGT\n")
    junk = tmp_path / "a.txt"; junk.write_text("hello\n")
    assert mc.auto_detect_file_type(str(fq)) == "fastq"
    assert mc.auto_detect_file_type(str(fa)) == "fasta"
    with pytest.raises(SystemExit):
        mc.auto_detect_file_type(str(junk))


def test_parse_seqs_fasta_fastq_multiline(tmp_path):
    p = tmp_path / "x.fq"
    p.write_text("@a desc\nACGT\nAC\n+\nIIII\nII\n@b\nGG\n+\nII\n")
    recs = list(mc.parse_seqs(open(str(p))))
    assert [(r.id, r.seq, r.quality) for r in recs] == [("a", "ACGTAC", "IIIIII"), ("b", "GG", "II")]
    p2 = tmp_path / "x.fa"
    p2.write_text(">s1 x\nAC\nGT\n>s2\nTT\n")
    assert [(r.id, r.seq) for r in mc.parse_seqs(open(str(p2)))] == [("s1", "ACGT"), ("s2", "TT")]


def test_imputed_arguments_match_reference_config1():
    g = golden("config1_example_fq")
    args = {"seqfiles": [os.path.join(INPUTS, "example.fq.gz")], "nreads": 10000, "read_length": 100, "threads": 1}
    mc.impute_missing_args(args)
    for k in ("file_type", "quality_offset", "read_length", "nreads", "min_quality", "mean_quality", "max_unknown", "filter_dups"):
        assert args[k] == g["args"][k], k


def test_auto_read_length_unittest_metagenome():
    args = {"seqfiles": [os.path.join(INPUTS, "metagenome.fa.gz")]}
    mc.impute_missing_args(args)
    assert args["read_length"] == 100 and args["file_type"] == "fasta" and args["nreads"] == 1000000


@pytest.mark.parametrize("case,args", [
    ("config1_example_fq", {"seqfiles": ["example.fq.gz"], "nreads": 10000, "read_length": 100, "threads": 1}),
    ("unittest_metagenome", {"seqfiles": ["metagenome.fa.gz"]}),
])
def test_process_seqfile_writes_the_same_reads(case, args, tmp_path):
    g = golden(case)
    args = dict(args, seqfiles=[os.path.join(INPUTS, f) for f in args["seqfiles"]])
    mc.impute_missing_args(args)
    paths = {"tempfile": str(tmp_path / "reads.fa")}
    mc.process_seqfile(args, paths)
    assert args["sampled_reads"] == g["sampled_reads"]
    assert hashlib.md5(open(paths["tempfile"], "rb").read()).hexdigest() == g["reads_md5"]


def test_quality_and_duplicate_filters(tmp_path):
    p = tmp_path / "q.fq"
    good = "ACGTACGTAC" * 6
    recs = [("r0", good, "I" * 60), ("short", "ACGT", "IIII"), ("lowq", good, "I" * 30 + "#" + "I" * 29), ("dup", good, "I" * 60),
            ("rc", mc.Sequence("x", good).reverse_complement(), "I" * 60), ("nn", "N" * 60, "I" * 60), ("r1", "TTGCA" * 12, "I" * 60)]
    p.write_text("".join("@%s\n%s\n+\n%s\n" % r for r in recs))
    args = {"seqfiles": [str(p)], "read_length": 50, "nreads": 100, "min_quality": 20, "mean_quality": 20, "filter_dups": True, "max_unknown": 10}
    mc.impute_missing_args(args)
    paths = {"tempfile": str(tmp_path / "out.fa")}
    mc.process_seqfile(args, paths)
    out = open(paths["tempfile"]).read().split("\n")
    assert args["sampled_reads"] == 2 and out[1] == good[:50] and out[3] == ("TTGCA" * 12)[:50]


@pytest.mark.parametrize("case", ["config1_example_fq", "unittest_metagenome"])
def test_classify_aggregate_estimate_match_reference(case, tmp_path):
    """The host restatement of classify_reads/aggregate_hits/estimate_average_genome_size on the reference's own
    m8 gives the reference's best_hits, per-family aggregates and AGS bit for bit."""
    g = golden(case)
    m8 = tmp_path / "t.m8"
    m8.write_bytes(gzip.open(os.path.join(GOLD, case + ".m8.gz"), "rb").read())
    args = {"read_length": g["args"]["read_length"], "verbose": False, "sampled_reads": g["sampled_reads"]}
    paths = {"tempfile": str(tmp_path / "t"), "db": None}
    best = mc.classify_reads(args, paths)
    assert best == g["best_hits"]
    assert list(best.keys()) == sorted(best.keys(), key=int)   # m8 order = ascending read id (the golden JSON is key-sorted)
    agg = mc.aggregate_hits(args, paths, best)
    assert agg == g["agg_hits"]
    assert mc.estimate_average_genome_size(args, paths, agg) == g["est_ags"]


def test_report_format(tmp_path):
    args = {"outfile": str(tmp_path / "o.txt"), "seqfiles": ["a", "b"], "sampled_reads": 5, "read_length": 100, "min_quality": -5, "mean_quality": -5,
            "filter_dups": False, "max_unknown": 100}
    mc.report_results(args, 3051745.7641809303, 980306)
    assert open(args["outfile"]).read() == ("Parameters\nmetagenome:\ta,b\nreads_sampled:\t5\ntrimmed_length:\t100\nmin_quality:\t-5\nmean_quality:\t-5\n"
                                            "filter_dups:\tFalse\nmax_unknown:\t100\n\nResults\naverage_genome_size:\t3051745.7641809303\ntotal_bases:\t980306\n"
                                            "genome_equivalents:\t0.32122793828571367\n")


def test_count_bases_config1():
    g = golden("config1_example_fq")
    assert mc.count_bases({"seqfiles": [os.path.join(INPUTS, "example.fq.gz")], "verbose": False}) == g["total_bases"]


def test_rapsearch_hook_runs_the_external_executable(ref_dir):
    """args['rapsearch'] (the reference's -r, microbe_census.py:110-111): the bundled RAPsearch2 binary runs as a subprocess on the
    database the library's own writer produces, the m8 it leaves is classified by the Python statement of classify_reads - the
    whole host side without a GPU: same AGS as the reference, bit for bit (BASELINE configs[0])."""
    import contextlib
    import io
    import json
    import os
    from microbecensus_amd import microbe_census as mc
    here = os.path.dirname(os.path.abspath(__file__))
    g = json.load(open(os.path.join(here, "golden", "config1_example_fq.json")))
    args = {"seqfiles": [os.path.join(here, "golden", "inputs", "example.fq.gz")], "nreads": 10000, "read_length": 100, "threads": 8,
            "rapsearch": os.path.join(ref_dir, "rapsearch_Linux_2.15"), "verbose": True}
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        res = mc.run_pipeline(args)
    assert res is not None, buf.getvalue()
    est, args = res
    assert args["sampled_reads"] == g["sampled_reads"] and est == g["est_ags"]
    assert "reads hit marker proteins" in buf.getvalue()


def test_rapsearch_hook_rejects_other_programs(tmp_path):
    import pytest
    from microbecensus_amd import microbe_census as mc
    fake = tmp_path / "fake"
    fake.write_text("#!/bin/sh\necho line1 >&2\necho 'not rapsearch' >&2\n")
    fake.chmod(0o755)
    with pytest.raises(SystemExit) as e:
        mc.check_rapsearch(str(fake))
    assert "Incorrect version of rapsearch2" in str(e.value)


def test_index_cache_directory_must_be_private(tmp_path, monkeypatch):
    """MC_INDEX_CACHE=<dir> passes the same test as the default per-user directory: this user's, not group / world
    writable - else nothing is cached."""
    from microbecensus_amd import _native
    good = tmp_path / "mine"
    assert _native._private_dir(str(good)) == str(good) and (os.stat(good).st_mode & 0o777) == 0o700
    bad = tmp_path / "shared"
    bad.mkdir()
    os.chmod(bad, 0o777)
    assert _native._private_dir(str(bad)) is None
    seen = []

    class Lib:
        def mc_set_index_cache(self, d):
            seen.append(d)
    monkeypatch.setattr(_native, "load_library", lambda: Lib())
    for d, want in ((bad, []), (good, [str(good).encode()])):
        monkeypatch.setattr(_native, "_index_cache_set", False)
        monkeypatch.setenv("MC_INDEX_CACHE", str(d))
        del seen[:]
        _native.use_index_cache()
        assert seen == want


def test_bench_line_models_price_asks_at_the_calibrated_rates():
    """bench.py's two line models (roofline.scattered_line_ceiling / extension_kernel_scattered_line_ceiling): asks x 1 / rate of the committed
    calibration, summed, over the measured duration - checked on made-up counts against the arithmetic done by hand."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cal = bench.gather_ceiling()
    assert cal is not None and cal["rate"][("16MB", 32)] > 0 and cal["rate"][("128MB", 32)] > 0
    n, hits, surv = 1_000_000, 75_000_000, 20_000_000
    e = bench.eval_line_model(cal, n, hits, surv, 2.0, 2.4)
    want = hits / (cal["rate"][("128MB", 32)] * 1e9) * 1e3 + 2.0 * surv / (cal["rate"][("16MB", 32)] * 1e9) * 1e3
    assert abs(e["model_ms_per_launch"] - want) < 2e-3 and abs(e["frac"] - want / 2.0) < 2e-3 and e["hits_per_read"] == 75.0
    asks = {"seed_exact_asks": 115 * n, "seed_wild_asks": 150 * n, "seed_pair_asks": 77 * n, "seed_probes": 42 * n}
    s = bench.seed_line_model(cal, 150, n, asks, hits, 5.4)
    assert s is not None and 0.5 < s["frac"] < 1.5 and abs(sum(r["ms_per_launch"] for r in s["by_structure"]) - s["model_ms_per_launch"]) < 5e-3
    assert bench.eval_line_model(None, n, hits, surv, 2.0, None) is None and bench.eval_line_model(cal, n, 0, 0, 2.0, None) is None
