"""GPU parity: the HIP pipeline (through the C ABI) against the golden m8 captured from the reference's
RAPsearch2 binary, and against the oracle restatement on the same reads."""
import gzip
import hashlib
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
pytestmark = pytest.mark.gpu


def golden_reads(case):
    meta = json.load(open(os.path.join(GOLD, case + ".json")))
    L = meta["args"]["read_length"]
    reads_gz = os.path.join(GOLD, case + ".reads.fa.gz")
    if os.path.exists(reads_gz):
        seqs = [l.strip() for l in gzip.open(reads_gz, "rt") if not l.startswith(">")]
    else:
        recs, seq = [], None
        with gzip.open(os.path.join(GOLD, "inputs", meta["seqfiles"][0]), "rt") as f:
            for line in f:
                if line[0] == ">":
                    if seq is not None:
                        recs.append("".join(seq))
                    seq = []
                else:
                    seq.append(line.strip())
            recs.append("".join(seq))
        seqs = [s[:L] for s in recs if len(s) >= L]
    arr = np.frombuffer("".join(seqs).encode(), dtype=np.uint8).reshape(len(seqs), L)
    return arr, meta


@pytest.fixture(scope="module")
def engine():
    from microbecensus_amd._native import Engine
    e = Engine(device=0)
    yield e
    e.close()


@pytest.mark.parametrize("counting", [False, True], ids=["filtered", "counting"])
@pytest.mark.parametrize("case", ["config1_example_fq", "unittest_metagenome"])
def test_m8_identical_to_reference(case, counting, engine, tmp_path):
    """Both forms of the seed kernel: the default one (10-mer Bloom filter in front of the range search) and the one
    that also counts the reference algorithm's index reads (no filter: every probe is searched)."""
    reads, meta = golden_reads(case)
    engine.set_run(reads.shape[1])
    engine.set_counting(counting)
    try:
        rows, _ = engine.search(reads)
    finally:
        engine.set_counting(False)
    out = str(tmp_path / "out.m8")
    engine.write_m8(out)
    got = open(out, "rb").read()
    st = engine.stats()
    print(case, st)
    assert len(rows) == meta["m8_rows"]
    assert hashlib.md5(got).hexdigest() == meta["m8_md5"]
    assert (st["bucket_lookups"] > 0 and st["key_probes"] > 0) if counting else (st["bucket_lookups"] == 0 and st["key_probes"] == 0)


_DIRTY_WORKER = r"""
import gzip, hashlib, json, os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
from microbecensus_amd import _native
gold = os.path.join(sys.argv[1], "tests", "golden")
meta = json.load(open(os.path.join(gold, "dirty_reads.json")))
seqs = [l.rstrip(b"\r\n") for l in gzip.open(os.path.join(gold, "dirty_reads.fa.gz"), "rb") if not l.startswith(b">")]
reads = np.frombuffer(b"".join(seqs), dtype=np.uint8).reshape(len(seqs), meta["read_length"])
eng = _native.Engine(device=0)
eng.set_run(meta["read_length"])
res = {}
for counting in (False, True):
    eng.set_counting(counting)
    rows, _ = eng.search(reads)
    eng.write_m8(sys.argv[2])
    res["counting" if counting else "filtered"] = [len(rows), hashlib.md5(open(sys.argv[2], "rb").read()).hexdigest()]
eng.close()
json.dump(res, open(sys.argv[3], "w"))
"""


@pytest.mark.parametrize("staged", ["1", "0"], ids=["reads_staged_in_lds", "reads_from_global"])
def test_dirty_reads_identical_to_reference(staged, tmp_path):
    """Reads that are not clean upper-case ACGT - lower case, IUPAC codes, `*`, `-`, `.`, digits, blanks, `?|~` - as the
    reference's binary searched them (tests/golden/make_dirty_golden.py; byte tables of `CHashSearch` ctor 0x4169bd-0x416a62,
    `BuildQHash@0x40b530`): the wave-level base decoding (`mc_nt_code`, packed 4-bit frames of the seed kernel) must give the
    same m8, for both forms of k_translate_seg (MC_TS_STAGED is read once per process: one child process per form)."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    meta = json.load(open(os.path.join(GOLD, "dirty_reads.json")))
    w = tmp_path / "w.py"
    w.write_text(_DIRTY_WORKER)
    env = dict(os.environ, MC_TS_STAGED=staged)
    subprocess.check_call([sys.executable, str(w), repo, str(tmp_path / "o.m8"), str(tmp_path / "o.json")], env=env, timeout=900)
    res = json.load(open(tmp_path / "o.json"))
    for form in ("filtered", "counting"):
        assert res[form] == [meta["m8_rows"], meta["m8_md5"]], form


def test_generic_seed_path_on_a_database_with_a_seed_threshold(tmp_path):
    """A second database whose `.info` threshold is 1 (6.5 M residues of random ORFs; golden from the reference's prerapsearch +
    rapsearch, tests/golden/make_generic_db_golden.py): mc_open() turns the position-parallel seed kernel off (it is exact only
    at threshold 0) and k_enumerate emits seeds of 6 .. 9 residues, which k_eval_seeds grows residue by residue
    (`Searching 0x4153a3-0x4153ce`, `0x415ec0-0x415f71`).  Through mc_open (FASTA) and mc_open_rapdb (the file mc_rapdb_write wrote)."""
    import sys
    sys.path.insert(0, GOLD)
    import make_generic_db_golden as G
    from microbecensus_amd import _native
    meta = json.load(open(os.path.join(GOLD, "generic_db.json")))
    names, seqs, reads = G.case_inputs()
    assert hashlib.md5(b"".join(b">%d\n%s\n" % (i, bytes(r)) for i, r in enumerate(reads))).hexdigest() == meta["reads_md5"]
    db = str(tmp_path / "db2")
    _native.rapdb_write(names, seqs, db)
    out = str(tmp_path / "out.m8")
    for how in ("fasta", "rapdb"):
        eng = _native.Engine(device=0, names=names, seqs=seqs, marker_family=[0] * len(names), nfam=1) if how == "fasta" else _native.Engine.from_rapdb(db, device=0, family_of={n: "all" for n in names}, families=["all"])
        try:
            eng.set_run(meta["read_length"])
            rows, _ = eng.search(reads)
            eng.write_m8(out)
            st = eng.stats()
        finally:
            eng.close()
        print(how, st)
        assert len(rows) == meta["m8_rows"] and hashlib.md5(open(out, "rb").read()).hexdigest() == meta["m8_md5"], how
        assert st["bucket_lookups"] > 0                  # (the generic kernel counts the reference algorithm's index reads: it is the one that ran)


@pytest.mark.parametrize("case", ["config1_example_fq", "dirty_reads"])
def test_every_stage_equals_the_emulation(case, engine, tmp_path):
    """Kernel by kernel (the candidates of every stage against the restatement's - what localises a difference the
    end-to-end m8 equality only reports): what the stages of mc_run_range leave on the device (mc_debug_stage) against what the
    CPU emulation of the same per-thread code makes (tests/emul/mc_emul, itself pinned on the reference's m8) -
    the six translated and SEG-masked frames of every read byte for byte (k_translate_seg; BuildQHash@0x40b530, Seg::*), the
    multiset of seed hits (k_enumerate_q; Searching@0x415050: written with the index of the posting, looked up here), the
    multiset of gap tasks and the multiset of HSPs, ungapped and gapped (k_eval_seeds, k_gap_*; ExtendSeq2Set@0x413b90,
    AlignGapped@0x40a550, CalRes@0x4077a0)."""
    import re
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    repo = os.path.dirname(here)
    exe = str(tmp_path / "mc_emul")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe, os.path.join(here, "emul", "mc_emul.cpp")])
    faa = tmp_path / "markers.faa"
    faa.write_bytes(gzip.open(os.path.join(repo, "microbecensus_amd", "data", "markers.faa.gz"), "rb").read())
    src = os.path.join(GOLD, "dirty_reads.fa.gz" if case == "dirty_reads" else case + ".reads.fa.gz")
    fa = tmp_path / "reads.fa"
    fa.write_bytes(gzip.open(src, "rb").read())
    pre = str(tmp_path / "emul")
    r = subprocess.run([exe, str(faa), str(fa), str(tmp_path / "e.m8")], stderr=subprocess.PIPE, check=True, env=dict(os.environ, MC_DUMP_STAGES=pre))
    err = r.stderr.decode()
    fp_e = int(re.search(r"stages dumped: FP (\d+)", err).group(1))
    seqs = [l.rstrip(b"\r\n") for l in gzip.open(src, "rb") if not l.startswith(b">")]
    reads = np.frombuffer(b"".join(seqs), dtype=np.uint8).reshape(len(seqs), len(seqs[0]))
    n, L = reads.shape
    engine.set_run(L)
    engine.upload(reads)
    engine.run_range(0, n)
    st = engine.stats()

    # ---- frames
    fe = np.fromfile(pre + ".frames", np.uint8).reshape(n, 6, fp_e)
    fg = engine.debug_stage(0)
    fg = fg.reshape(n, 6, fg.shape[1])
    INV = 20
    for f in range(6):
        k = (L - f % 3) // 3
        assert np.array_equal(fg[:, f, :k], fe[:, f, :k]), "frame %d" % f
        assert (fg[:, f, k:] == INV).all()
    k0 = L // 3
    assert (fe[:, 0, :k0] != INV).mean() > 0.5 and (fe[:, 0, :k0] == INV).any()        # (residues, and masked / stop positions among them)
    # ---- seed hits
    task_dt = np.dtype([("read", "<u4"), ("chrono", "<u4"), ("posting", "<u4"), ("w3", "<u4")])
    te = np.fromfile(pre + ".tasks", task_dt)
    tg = engine.debug_stage(1).reshape(-1).view(task_dt)
    tg = tg[tg["read"] != 0xFFFFFFFF]                                                   # (padding of the blocks of the pool)
    # the product's seed kernel (k_enumerate_q) writes the INDEX of a hit's posting in the index's posting array, not the posting (the
    # evaluation kernel fetches it with the subject's offsets in one load: MC_POST8)
    post = engine.index_view()["post"].astype(np.int64)
    assert len(tg) == len(te) == st["seed_tasks"] and len(te) > 1000
    assert (tg["read"] >> 21 == 0).all() and (tg["w3"] & 0xFFFFFF == 0).all()
    norm_e = np.stack([te["read"], te["chrono"], te["posting"], te["w3"] & 0xFF, te["w3"] >> 8], 1).astype(np.int64)
    norm_g = np.stack([tg["read"], tg["chrono"], post[tg["posting"].astype(np.int64)], (tg["w3"] >> 24) & 15, tg["w3"] >> 28], 1).astype(np.int64)
    assert np.array_equal(norm_e[np.lexsort(norm_e.T[::-1])], norm_g[np.lexsort(norm_g.T[::-1])])
    # ---- gap tasks
    gap_dt = np.dtype({"names": ["read", "chrono", "sidx", "qp", "dp", "L", "qfwd", "qbwd", "score", "nmatch"],
                       "formats": ["<u4", "<u4", "<u4", "<i2", "<i2", "<i2", "<i2", "<i2", "<i2", "<i2"], "offsets": [0, 4, 8, 12, 14, 16, 18, 20, 22, 24], "itemsize": 28})
    hsp_dt = np.dtype({"names": ["read", "chrono", "sidx", "score", "frame", "alnlen", "mism", "gaps", "nmatch", "qaas", "qaae", "ds", "de", "qnts", "qnte", "loge"],
                       "formats": ["<u4", "<u4", "<i4"] + ["<i2"] * 12 + ["<f8"], "offsets": [0, 4, 8] + list(range(12, 36, 2)) + [40], "itemsize": 48})

    def table(a):                                                                       # the named fields (no padding bytes), rows in a canonical order
        t = np.stack([a[k].astype(np.float64) for k in a.dtype.names], 1)
        return t[np.lexsort(t.T[::-1])]
    ge = np.fromfile(pre + ".gaps", gap_dt)
    gg = engine.debug_stage(2).reshape(-1).view(gap_dt)
    gg = gg[gg["read"] != 0xFFFFFFFF]
    assert len(gg) == len(ge) == st["gap_tasks"] and len(ge) > 100
    assert np.array_equal(table(ge), table(gg))
    # ---- HSPs
    he = np.fromfile(pre + ".hsps", hsp_dt)
    hg = engine.debug_stage(3).reshape(-1).view(hsp_dt)
    hg = hg[hg["read"] != 0xFFFFFFFF]
    assert len(hg) == len(he) == st["hsps"] and len(he) > 1000 and (he["gaps"] > 0).any()
    assert np.array_equal(table(he), table(hg))


def test_the_product_library_is_the_one_loaded():
    """The GPU tests must exercise microbecensus_amd/libmcensus_hip.so itself: no MCENSUS_LIB override in the environment, and
    the library mapped into this process is the in-tree one."""
    from microbecensus_amd import _native
    assert "MCENSUS_LIB" not in os.environ
    _native.load_library()
    want = os.path.realpath(os.path.join(os.path.dirname(os.path.abspath(_native.__file__)), "libmcensus_hip.so"))
    mapped = {os.path.realpath(line.split()[-1]) for line in open("/proc/self/maps") if "libmcensus_hip" in line}
    assert mapped == {want}
