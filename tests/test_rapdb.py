"""Reader of the database format prerapsearch writes (boost binary archive): mc_open_rapdb / mc_rapdb_verify."""
import gzip
import hashlib
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
RAPDB = os.path.join(REPO, "oracle", "_ref", "rapdb_2.15")
needs_db = pytest.mark.skipif(not os.path.exists(RAPDB), reason="oracle/_ref/rapdb_2.15 (built from the reference's prerapsearch) not present")


@needs_db
def test_rapdb_is_the_index_we_build_from_the_fasta():
    """No GPU: the file prerapsearch_Linux_2.15 wrote holds exactly the residues, buckets, posting order and suffix keys
    that mc_open() derives from the marker FASTA."""
    from microbecensus_amd import _native
    names, seqs = _native.load_markers()
    assert _native.rapdb_verify(RAPDB, names, seqs) == 0
    # and it notices a difference: swap two residues of one marker
    seqs2 = list(seqs)
    s = seqs2[10]
    seqs2[10] = s[:20] + s[21] + s[20] + s[22:]
    if seqs2[10] != s:
        assert _native.rapdb_verify(RAPDB, names, seqs2) > 0


def test_rapdb_reader_rejects_garbage(tmp_path):
    from microbecensus_amd import _native
    p = tmp_path / "x.db"
    p.write_bytes(b"not an archive" * 100)
    assert _native.rapdb_verify(str(p), ["a"], ["ACDEFGHIKLMNPQRSTVWY"]) < 0
    assert b"archive" in _native.load_library().mc_last_error()


@needs_db
@pytest.mark.gpu
def test_engine_on_the_prerapsearch_database():
    """The engine opened on the prerapsearch database gives the reference's m8, byte for byte (config 1)."""
    from microbecensus_amd import _native
    meta = json.load(open(os.path.join(HERE, "golden", "config1_example_fq.json")))
    L = meta["args"]["read_length"]
    seqs = [l.strip() for l in gzip.open(os.path.join(HERE, "golden", "config1_example_fq.reads.fa.gz"), "rt") if not l.startswith(">")]
    reads = np.frombuffer("".join(seqs).encode(), dtype=np.uint8).reshape(len(seqs), L)
    eng = _native.Engine.from_rapdb(RAPDB, device=0)
    try:
        model = _native.load_model()
        eng.set_run(L, model["pars"][str(L)], model["families"])
        rows, best = eng.search(reads)
        out = os.path.join(os.environ.get("TMPDIR", "/tmp"), "rapdb_engine.m8")
        eng.write_m8(out)
        assert hashlib.md5(open(out, "rb").read()).hexdigest() == meta["m8_md5"]
        assert len(best) == len(meta["best_hits"])
    finally:
        eng.close()


@needs_db
def test_writer_reproduces_prerapsearch_byte_for_byte(tmp_path):
    """mc_rapdb_write (index built from the FASTA, written in RAPSearch2 2.15's on-disk format) == the two files
    `prerapsearch_Linux_2.15 -d markers.faa -n rapdb_2.15` wrote, byte for byte: bucket order, suffix keys, .info median and
    letter frequencies included."""
    from microbecensus_amd import _native
    names, seqs = _native.load_markers()
    out = str(tmp_path / "db")
    _native.rapdb_write(names, seqs, out)
    for suffix in ("", ".info"):
        assert hashlib.md5(open(out + suffix, "rb").read()).hexdigest() == hashlib.md5(open(RAPDB + suffix, "rb").read()).hexdigest(), suffix


def test_index_cache_round_trip(tmp_path):
    """The per-user cache of built indexes (mc_set_index_cache; DESIGN 3, cold start): what mc_open() would read back is, array by
    array, what mc_build_index built; a file written for other sequences, a file with one flipped byte and a truncated file are
    refused (and the index rebuilt).  No GPU."""
    import ctypes as C
    from microbecensus_amd import _native
    names, seqs = _native.load_markers()
    lib = _native.load_library()
    n = len(names)
    rc = lib.mc_index_cache_check((C.c_char_p * n)(*[s.encode() for s in names]), (C.c_char_p * n)(*[s.encode() for s in seqs]), n, str(tmp_path).encode())
    assert rc == 0, lib.mc_last_error().decode()
    assert not list(tmp_path.iterdir())
