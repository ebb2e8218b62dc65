"""GPU parity where the five goldens and the error-free genome reads do not reach: libraries dense in marker genes (the pools
sized for shotgun reads overflow and mc_run_range runs the range again in halves), and reads with sequencing errors
(substitutions and indels: frame shifts, gapped DP band growth).  All calls go through the C ABI."""
import os

import numpy as np
import pytest

from test_gpu_pipeline import _oracle_rows, _rows, assert_rows_equal

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def engine():
    from microbecensus_amd._native import Engine
    e = Engine(device=0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def marker_reads():
    """300,000 reads of 150 bp from a synthetic community made of nothing but (diverged) marker genes: ~290 HSPs and ~100 m8
    rows per read where a shotgun library has 23 and 1.9."""
    from microbecensus_amd import _native, synth
    names, seqs = _native.load_markers()
    genome = synth.build_genomes(seqs, total_bp=3_000_000, seed=404, marker_gene_fraction=1.0)
    return synth.sample_reads(genome, 300_000, 150, seed=405)


def _fields_equal(a, b):
    return len(a) == len(b) and all(np.array_equal(a[f], b[f]) for f in a.dtype.names)


def test_pool_overflow_runs_the_range_in_halves(engine, marker_reads, monkeypatch):
    """ONE mc_search of a marker-dense library: every batch overflows the HSP / gap-task pools (`C_OVERFLOW`, -2 out of
    run_range_once) and mc_run_range answers by running the range in halves, and those in halves again.  The rows must be the
    ones the same reads give in 5,000-read batches (which fit), and the oracle's on a subsample."""
    from microbecensus_amd import _native
    model = _native.load_model()
    fams = model["families"]
    engine.set_run(150, model["pars"]["150"], fams)
    rows, best = engine.search(marker_reads)
    st = engine.stats()
    print("one call:", st)
    assert st["range_splits"] > 0, "the batch did not overflow: the test no longer exercises the halving path"
    assert st["reads"] == len(marker_reads) and st["rows"] == len(rows) and st["classified"] == len(best)
    assert np.all(np.diff(rows["query"]) >= 0) and np.bincount(rows["query"]).max() <= 500
    monkeypatch.setenv("MC_STREAM_BATCH", "5000")
    rows5, best5 = engine.search(marker_reads)
    st5 = engine.stats()
    monkeypatch.delenv("MC_STREAM_BATCH")
    print("5,000-read batches:", st5)
    assert st5["range_splits"] == 0
    assert _fields_equal(rows, rows5) and _fields_equal(best, best5)
    assert (st["hsps"], st["gap_tasks"], st["seed_tasks"]) == (st5["hsps"], st5["gap_tasks"], st5["seed_tasks"])
    sub = rows[rows["query"] < 5000]
    assert_rows_equal(_rows(sub), _oracle_rows(marker_reads[:5000]))


def test_pool_overflow_through_the_file_pipeline_best_hits_only(engine, marker_reads, tmp_path):
    """The same library as a FASTA file through mc_search_files with mc_set_best_hits_only (what run_pipeline runs): best hits ==
    those of the rows path above's algorithm (searched again here), with the halving path taken."""
    from microbecensus_amd import _native
    model = _native.load_model()
    fams = model["families"]
    fa = tmp_path / "markers_only.fa"
    with open(fa, "wb") as f:
        for s in range(0, len(marker_reads), 50000):
            f.write(b"".join(b">r%d\n%s\n" % (s + i, bytes(r)) for i, r in enumerate(marker_reads[s:s + 50000])))
    engine.set_run(150, model["pars"]["150"], fams)
    rd = _native.Reader([str(fa)], 150, 10_000_000, False, 0, -5, -5, 100, False)
    try:
        _, best = engine.search_files(rd, keep_rows=False, best_only=True)
        st = engine.stats()
        assert rd.stats()["sampled"] == len(marker_reads)
    finally:
        rd.close()
    print("files, best hits only:", st)
    assert st["range_splits"] > 0
    _, want = engine.search(marker_reads)
    assert len(best) > 1000 and _fields_equal(best, want)


@pytest.mark.parametrize("L,n,sub,indel", [(150, 6000, 0.01, 0.002), (150, 6000, 0.05, 0.01), (300, 3000, 0.02, 0.005), (300, 3000, 0.05, 0.01), (500, 1500, 0.03, 0.01)])
def test_reads_with_sequencing_errors_against_oracle(engine, L, n, sub, indel):
    """1 - 5 % substitutions and 0.2 - 1 % indels at 150 / 300 / 500 bp (synth.mutate_reads): mismatching seeds, frame shifts in
    the middle of a hit, gapped extensions whose band has to grow (`AlignGapped@0x40a550`) - every m8 row identical to the
    oracle's."""
    from microbecensus_amd import _native, synth
    names, seqs = _native.load_markers()
    genome = synth.build_genomes(seqs, total_bp=600_000, seed=900 + L, marker_gene_fraction=0.3)
    clean = synth.sample_reads(genome, n, L + 24, seed=L + 7)
    reads = synth.mutate_reads(clean, L, sub_rate=sub, indel_rate=indel, seed=L + int(sub * 1000))
    assert (reads != clean[:, :L]).any(axis=1).mean() > 0.5
    engine.set_run(L)
    rows, _ = engine.search(reads)
    got = _rows(rows)
    assert_rows_equal(got, _oracle_rows(reads))
    assert len(rows) > 100 and sum(1 for r in got if r[4] > 0) > 5          # (rows with gap openings among them)


@pytest.mark.parametrize("thr", [-3.0, 0.0, 2.5])
def test_log_e_threshold_other_than_one_against_oracle(engine, thr, tmp_path):
    """`-e` (mc_set_run's loge_thr; scripts/rapsearch_mi355x forwards it): the reference runs `-e 1`, RAPsearch2 takes any threshold - it
    decides which single HSPs are kept at all (`CalRes 0x4078e4-0x4078f9`), which chains of sum statistics stay (`SumEvalue`), where
    `PrintRes` stops - and with it which reads the ordering kernels mark.  m8 text of the HIP path == the oracle's (RS_LOGE_THR)."""
    import hashlib
    import subprocess
    from microbecensus_amd import _native, synth
    names, seqs = _native.load_markers()
    genome = synth.build_genomes(seqs, total_bp=500_000, seed=71, marker_gene_fraction=0.25)
    reads = synth.sample_reads(genome, 5000, 150, seed=72)
    fa = tmp_path / "r.fa"
    fa.write_bytes(b"".join(b">%d\n%s\n" % (i, bytes(r)) for i, r in enumerate(reads)))
    engine.set_run(150, loge_thr=thr)
    try:
        rows, _ = engine.search(reads)
        engine.write_m8(str(tmp_path / "gpu.m8"))
    finally:
        engine.set_run(150)
    subprocess.check_call([os.path.join(REPO, "oracle", "rs_port"), os.path.join(REPO, "oracle", "_ref", "rapdb_2.15"), str(fa), str(tmp_path / "cpu.m8")],
                          env=dict(os.environ, RS_LOGE_THR=repr(thr)))
    got, want = (tmp_path / "gpu.m8").read_bytes(), (tmp_path / "cpu.m8").read_bytes()
    assert want.count(b"\n") > 50 and hashlib.md5(got).hexdigest() == hashlib.md5(want).hexdigest(), (len(rows), want.count(b"\n"))


_SMALL_WORKER = r"""
import gzip, hashlib, json, os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
from microbecensus_amd import _native, synth
gold = os.path.join(sys.argv[1], "tests", "golden")
out = {}
eng = _native.Engine(device=0)
for case, L in (("config1_example_fq", 100), ("unittest_metagenome", 100)):
    meta = json.load(open(os.path.join(gold, case + ".json")))
    if os.path.exists(os.path.join(gold, case + ".reads.fa.gz")):
        seqs = [l.strip() for l in gzip.open(os.path.join(gold, case + ".reads.fa.gz"), "rb") if not l.startswith(b">")]
    else:
        recs, seq = [], None
        for line in gzip.open(os.path.join(gold, "inputs", meta["seqfiles"][0]), "rb"):
            if line[:1] == b">":
                if seq is not None: recs.append(b"".join(seq))
                seq = []
            else:
                seq.append(line.strip())
        recs.append(b"".join(seq))
        seqs = [s[:L] for s in recs if len(s) >= L]
    reads = np.frombuffer(b"".join(seqs), dtype=np.uint8).reshape(len(seqs), L)
    eng.set_run(L)
    rows, _ = eng.search(reads)
    eng.write_m8(sys.argv[2])
    out[case] = [len(rows), hashlib.md5(open(sys.argv[2], "rb").read()).hexdigest(), meta["m8_rows"], meta["m8_md5"]]
# a read of the genome set with ~200 HSPs nearly all of which print, among copies: the wave kernels and the heap sort
read = np.frombuffer(b"CCTGGCAATGATGACTCCATCAGAGCAATTGGTTATTACGCAAGAGAAAT", dtype=np.uint8)
eng.set_run(50)
rows, _ = eng.search(np.tile(read, (70, 1)))
out["heavy_read_rows"] = len(rows)
json.dump(out, open(sys.argv[3], "w"))
"""


def test_ordering_paths_of_the_longest_reads_on_ordinary_reads(tmp_path):
    """The ordering kernels treat segments by size: up to 32 HSPs ranked by counting, up to 512 merge-sorted by a wave, up to 2048 /
    8192 by a workgroup, longer ones in blocks merged in global memory (k_order.h) - on the test sets nothing is longer than a few
    thousand HSPs.  Here the library is built with arrays of 64 / 128 / 256 items, so that ordinary reads of marker genes take
    every one of those paths, the merge in global memory included - and with finishing kernels of 128 / 192 / 256 stacked HSPs, so
    that the largest reads are finished by lane 0 in global scratch (k_finish_heavy's last resort): the reference's m8, byte for byte."""
    import json
    import subprocess
    import sys
    csrc = os.path.join(REPO, "microbecensus_amd", "csrc")
    lib = str(tmp_path / "libsmall.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                           "-DMC_ORDER_SMALL=64", "-DMC_ORDER_MID=128", "-DMC_ORDER_LDS=256", "-DMC_FH_N1=128", "-DMC_FH_N2=192", "-DMC_FH_N3=256", "-o", lib,
                           os.path.join(csrc, "mc_hip.hip"), os.path.join(csrc, "mc_reader.cpp"), "-lz", "-ldl", "-pthread"], timeout=900)
    w = tmp_path / "w.py"
    w.write_text(_SMALL_WORKER)
    subprocess.check_call([sys.executable, str(w), REPO, str(tmp_path / "o.m8"), str(tmp_path / "o.json")], env=dict(os.environ, MCENSUS_LIB=lib), timeout=900)
    res = json.load(open(tmp_path / "o.json"))
    for case in ("config1_example_fq", "unittest_metagenome"):
        assert res[case][:2] == res[case][2:], case
    assert res["heavy_read_rows"] == 198 * 70


def test_stream_of_ranges_equals_ranges_one_at_a_time(engine, marker_reads, monkeypatch):
    """mc_range_begin / mc_range_end (the front of range i + 1 enqueued - over the pools of range i - before the host looks at the
    results of range i) against mc_run_range on the same resident reads: rows, best hits and counts equal, range by range, in the
    order end, begin, results; the misuse errors; a range whose pools overflow
    comes back with -2 from mc_range_end and leaves nothing in flight; and the streaming call that meets such ranges (mc_search
    of the marker-dense library in 150,000-read batches) still delivers every row in order."""
    from microbecensus_amd import _native, synth
    names, seqs = _native.load_markers()
    model = _native.load_model()
    genome = synth.build_genomes(seqs, total_bp=2_000_000, seed=21)
    reads = synth.sample_reads(genome, 240_000, 150, seed=22)
    engine.set_run(150, model["pars"]["150"], model["families"])
    engine.upload(reads)
    n, step = len(reads), 30_000
    want = []
    for at in range(0, n, step):
        engine.run_range(at, step, at)
        want.append((engine.rows(), engine.best_hits(), engine.stats()))
    assert sum(len(w[0]) for w in want) > 100_000
    starts = list(range(0, n, step))
    got = []
    engine.range_begin(starts[0], step, starts[0])
    with pytest.raises(RuntimeError, match="in flight already"):
        engine.range_begin(0, step, 0)
    with pytest.raises(RuntimeError, match="in flight"):
        engine.run_range(0, step, 0)
    for i in range(len(starts)):
        nxt = starts[i + 1] if i + 1 < len(starts) else None
        engine.range_end()
        if nxt is not None:
            engine.range_begin(nxt, step, nxt)
            assert engine.ranges_in_flight() == 1
        got.append((engine.rows(), engine.best_hits(), engine.stats()))        # (while the front of the next range overwrites the pools)
    assert engine.ranges_in_flight() == 0
    with pytest.raises(RuntimeError, match="no range in flight"):
        engine.range_end()
    assert len(got) == len(want)
    for (r, b, s), (r0, b0, s0) in zip(got, want):
        assert _fields_equal(r, r0) and _fields_equal(b, b0)
        assert all(s[k] == s0[k] for k in ("reads", "seed_tasks", "gap_tasks", "hsps", "rows", "reads_with_rows", "classified"))
    # the streaming call runs its batches the same way
    monkeypatch.setenv("MC_STREAM_BATCH", "20000")
    rows_s, best_s = engine.search(reads)
    assert _fields_equal(rows_s, np.concatenate([w[0] for w in want])) and _fields_equal(best_s, np.concatenate([w[1] for w in want]))
    # overflowing ranges in a stream: every batch of 150,000 marker-dense reads overflows the pools
    monkeypatch.setenv("MC_STREAM_BATCH", "5000")
    rows5, best5 = engine.search(marker_reads)
    assert engine.stats()["range_splits"] == 0
    monkeypatch.setenv("MC_STREAM_BATCH", "150000")
    rows_o, best_o = engine.search(marker_reads)
    st = engine.stats()
    assert st["range_splits"] > 0 and st["reads"] == len(marker_reads) and _fields_equal(rows_o, rows5) and _fields_equal(best_o, best5)
    engine.upload(marker_reads)
    engine.range_begin(0, 150_000, 0)
    with pytest.raises(RuntimeError, match=r"\(-2\)"):
        engine.range_end()
    assert engine.ranges_in_flight() == 0
    engine.run_range(0, 150_000, 0)
    assert engine.stats()["range_splits"] > 0 and _fields_equal(engine.rows(), rows5[rows5["query"] < 150_000])


def test_stream_whose_batches_outgrow_the_pools(tmp_path):
    """A stream of unknown length (`-n` beyond anything a file holds: mc_search_files cannot size the pools ahead) on a fresh engine:
    the batches grow (256 k reads, then 512 k), the pools are replaced at the front of the second batch - while the best hits of the
    first one still lie in the pinned buffer of the old pools.  Best hits == those of the same reads searched from memory."""
    from microbecensus_amd import _native, synth
    names, seqs = _native.load_markers()
    model = _native.load_model()
    genome = synth.build_genomes(seqs, total_bp=2_000_000, seed=31)
    reads = synth.sample_reads(genome, 600_000, 100, seed=32)
    fa = tmp_path / "reads.fa"
    with open(fa, "wb") as f:
        for s in range(0, len(reads), 50000):
            f.write(b"".join(b">r%d\n%s\n" % (s + i, bytes(r)) for i, r in enumerate(reads[s:s + 50000])))
    eng = _native.Engine(device=0)
    try:
        eng.set_run(100, model["pars"]["100"], model["families"])
        rd = _native.Reader([str(fa)], 100, 1 << 42, False, 0, -5, -5, 100, False)
        try:
            _, best = eng.search_files(rd, keep_rows=False, best_only=True)
            assert rd.stats()["sampled"] == len(reads)
        finally:
            rd.close()
        _, want = eng.search(reads)
        assert len(want) > 1000 and _fields_equal(best, want)
    finally:
        eng.close()
