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


def test_the_product_library_is_the_one_loaded():
    """The GPU tests must exercise microbecensus_amd/libmcensus_hip.so itself: no MCENSUS_LIB override in the environment, and
    the library mapped into this process is the in-tree one."""
    from microbecensus_amd import _native
    assert "MCENSUS_LIB" not in os.environ
    _native.load_library()
    want = os.path.realpath(os.path.join(os.path.dirname(os.path.abspath(_native.__file__)), "libmcensus_hip.so"))
    mapped = {os.path.realpath(line.split()[-1]) for line in open("/proc/self/maps") if "libmcensus_hip" in line}
    assert mapped == {want}
