"""Native read sampler (mc_reader_*, csrc/mc_reader.cpp) and its Python statement against golden vectors produced by the
reference's own process_seqfile / count_bases on small synthetic files (tests/golden/make_sampler_golden.py).  No GPU."""
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
CASES = json.load(open(os.path.join(GOLD, "sampler_cases.json")))["cases"]


def _args(case):
    a = {"seqfiles": [os.path.join(GOLD, "sampler", f) for f in case["files"]], "verbose": False}
    a.update(case["args"])
    return a


def _prepare(mc, case):
    args = _args(case)
    mc.check_input(args)
    mc.impute_missing_args(args)
    mc.check_arguments(args)
    return args


# the reader cuts its input into regions (carry of the unfinished record) and pieces (speculative parallel parse, stitched in
# order): the default sizes put these small files into one piece, the tiny ones make every record boundary a seam
GEOMETRIES = [None, (4096, 300), (700, 97), (64, 1)]


@pytest.mark.parametrize("geometry", GEOMETRIES, ids=["default", "r4096-p300", "r700-p97", "r64-p1"])
@pytest.mark.parametrize("case", CASES, ids=[c["case"] for c in CASES])
def test_native_reader_matches_reference(case, geometry, tmp_path, monkeypatch):
    from microbecensus_amd import _native, microbe_census as mc
    if geometry:
        monkeypatch.setenv("MC_READER_REGION_BYTES", str(geometry[0]))
        monkeypatch.setenv("MC_READER_PIECE_BYTES", str(geometry[1]))
        monkeypatch.setenv("MC_READER_THREADS", "4")
    args = _prepare(mc, case)
    out = str(tmp_path / "tmp.fa")
    call = lambda: _native.sample_reads(args["seqfiles"], args["read_length"], args["nreads"], args["file_type"] == "fastq",   # noqa: E731
                                        args.get("quality_offset") or 0, args["min_quality"], args["mean_quality"], args["max_unknown"],
                                        args["filter_dups"], out)
    if "raises" in case:
        with pytest.raises(_native.ReferenceError_) as e:
            call()
        assert case["raises"] in str(e.value)
    else:
        reads, st = call()
        if "exit" in case:
            assert st["sampled"] == 0
        else:
            assert st["sampled"] == case["sampled_reads"] and reads.shape == (case["sampled_reads"], args["read_length"])
            assert (st["too_short"], st["low_qual"], st["dups"]) == (case["too_short"], case["low_qual"], case["dups"])
            assert open(out).read() == case["temp_fasta"]
            assert b"".join(b">%d\n%s\n" % (i, bytes(r)) for i, r in enumerate(reads)).decode() == case["temp_fasta"]
    if "count_bases" in case:
        assert _native.count_bases(args["seqfiles"]) == case["count_bases"]


@pytest.mark.parametrize("case", CASES, ids=[c["case"] for c in CASES])
def test_process_seqfile_stage(case, tmp_path, capsys):
    """The stage function as run_pipeline calls it (native reader inside): same file, same counters, same exits."""
    from microbecensus_amd import microbe_census as mc
    args = _prepare(mc, case)
    args["verbose"] = True
    paths = {"tempfile": str(tmp_path / "tmp.fa")}
    if "raises" in case:
        with pytest.raises(Exception):
            mc.process_seqfile(args, paths)
        return
    if "exit" in case:
        with pytest.raises(SystemExit) as e:
            mc.process_seqfile(args, paths)
        assert str(e.value) == case["exit"]
        return
    mc.process_seqfile(args, paths)
    assert args["sampled_reads"] == case["sampled_reads"]
    assert open(paths["tempfile"]).read() == case["temp_fasta"]
    text = capsys.readouterr().out
    assert "\t%d reads shorter than %d bp and skipped" % (case["too_short"], args["read_length"]) in text
    assert "\t%d low quality reads found and skipped" % case["low_qual"] in text
    assert "\t%d duplicate reads found and skipped" % case["dups"] in text
    assert mc.count_bases(args) == case["count_bases"]


@pytest.mark.parametrize("case", [c for c in CASES if "raises" not in c and "exit" not in c], ids=lambda c: c["case"])
def test_python_statement_of_the_sampler(case, tmp_path):
    """_process_seqfile_py (used for .bz2 inputs) follows the same rules."""
    from microbecensus_amd import microbe_census as mc
    args = _prepare(mc, case)
    paths = {"tempfile": str(tmp_path / "tmp.fa")}
    reads, st = mc._process_seqfile_py(args, paths)
    assert (st["sampled"], st["too_short"], st["low_qual"], st["dups"]) == (case["sampled_reads"], case["too_short"], case["low_qual"], case["dups"])
    assert open(paths["tempfile"]).read() == case["temp_fasta"]


def test_reader_golden_inputs_of_the_reference():
    """The reference's own example / unit-test files: counters of BASELINE configs[0] and of the unit test."""
    from microbecensus_amd import _native
    reads, st = _native.sample_reads([os.path.join(GOLD, "inputs", "example.fq.gz")], 100, 10000, True, 32, -5, -5, 100, False)
    assert (st["sampled"], st["too_short"]) == (8672, 1328)
    assert _native.count_bases([os.path.join(GOLD, "inputs", "example.fq.gz")]) == 980306
    reads, st = _native.sample_reads([os.path.join(GOLD, "inputs", "metagenome.fa.gz")], 100, 1000000, False, 0, -5, -5, 100, False)
    assert st["sampled"] == 70623


def test_count_bases_comes_from_the_sampler_pass_when_it_saw_everything(tmp_path, monkeypatch):
    """process_seqfile that reads every record leaves count_bases() its answer (no second pass over the files); a sampler
    stopped by nreads does not."""
    from microbecensus_amd import _native, microbe_census as mc
    case = [c for c in CASES if c["case"] == "fa_default"][0]
    args = _prepare(mc, case)
    mc.process_seqfile(args, {"tempfile": str(tmp_path / "a.fa")})
    monkeypatch.setattr(_native, "count_bases", lambda paths: (_ for _ in ()).throw(AssertionError("second pass")))
    assert mc.count_bases(args) == case["count_bases"]
    monkeypatch.undo()
    case = [c for c in CASES if c["case"] == "fa_n10"][0]
    args = _prepare(mc, case)
    mc.process_seqfile(args, {"tempfile": str(tmp_path / "b.fa")})
    assert tuple(args["seqfiles"]) not in mc._bases_cache
    assert mc.count_bases(args) == case["count_bases"]


def test_truncated_gz_is_an_error_not_a_short_file(tmp_path):
    """gzip.open raises EOFError on a stream cut in half (reference open_file :47-59), run_pipeline prints it and returns None;
    the native reader must not hand out a partial sample or a partial count_bases()."""
    from microbecensus_amd import _native, microbe_census as mc
    src = os.path.join(GOLD, "inputs", "example.fq.gz")
    data = open(src, "rb").read()
    cut = str(tmp_path / "cut.fq.gz")
    with open(cut, "wb") as f:
        f.write(data[: len(data) // 2])
    with pytest.raises(_native.ReferenceError_) as e:
        _native.sample_reads([cut], 100, 1000000, True, 32, -5, -5, 100, False)
    assert "EOFError" in str(e.value)
    with pytest.raises(_native.ReferenceError_):
        _native.count_bases([cut])
    # a sampler that stops (nreads reached) before the damage never sees it - like the reference's generator
    reads, st = _native.sample_reads([cut], 100, 100, True, 32, -5, -5, 100, False)
    assert st["sampled"] == 100
    assert tuple([cut]) not in mc._bases_cache


def test_nreads_none_means_no_cap():
    from microbecensus_amd import _native
    reads, st = _native.sample_reads([os.path.join(GOLD, "sampler", "a.fa")], 50, None, False, 0, -5, -5, 100, False)
    assert st["sampled"] == 60 and st["exhausted"] == 1


def test_bz2_input_is_read_natively(tmp_path):
    """.bz2 goes through the same native sampler (libbz2 bound at run time): same reads as the .gz of the same text."""
    import bz2
    import gzip
    from microbecensus_amd import _native
    text = gzip.open(os.path.join(GOLD, "inputs", "example.fq.gz"), "rb").read()
    p = str(tmp_path / "example.fq.bz2")
    with open(p, "wb") as f:
        f.write(bz2.compress(text))
    a, sa = _native.sample_reads([os.path.join(GOLD, "inputs", "example.fq.gz")], 100, 10000, True, 32, -5, -5, 100, False)
    b, sb = _native.sample_reads([p], 100, 10000, True, 32, -5, -5, 100, False)
    assert sa == sb and (a == b).all() and sb["sampled"] == 8672
    assert _native.count_bases([p]) == 980306


def test_bz2_with_several_streams_is_read_to_its_end(tmp_path):
    """pbzip2 / lbzip2 / `cat a.bz2 b.bz2` write several bzip2 streams into one file; Python's bz2 module (what the
    reference's open_file :55-58 uses) reads all of them, and so does the native reader.  Bytes behind the last stream that are
    no stream are ignored like Python ignores them; a file cut inside a stream is an EOFError."""
    import bz2
    import gzip
    from microbecensus_amd import _native
    text = gzip.open(os.path.join(GOLD, "inputs", "example.fq.gz"), "rb").read()
    lines = text.split(b"\n")
    cut = (len(lines) // 8) * 4                                   # a record boundary near the middle
    a, b = b"\n".join(lines[:cut]) + b"\n", b"\n".join(lines[cut:])
    want, sw = _native.sample_reads([os.path.join(GOLD, "inputs", "example.fq.gz")], 100, 10000, True, 32, -5, -5, 100, False)
    for name, blob in (("two", bz2.compress(a) + bz2.compress(b)), ("three", bz2.compress(a[:1000]) + bz2.compress(a[1000:]) + bz2.compress(b)),
                       ("trailing", bz2.compress(a) + bz2.compress(b) + b"not a stream")):
        p = str(tmp_path / (name + ".fq.bz2"))
        with open(p, "wb") as f:
            f.write(blob)
        assert bz2.open(p).read() == text
        got, sg = _native.sample_reads([p], 100, 10000, True, 32, -5, -5, 100, False)
        assert sg == sw and (got == want).all(), name
        assert _native.count_bases([p]) == 980306, name
    p = str(tmp_path / "cut.fq.bz2")
    with open(p, "wb") as f:
        f.write((bz2.compress(a) + bz2.compress(b))[:-40])
    with pytest.raises(_native.ReferenceError_) as e:
        _native.count_bases([p])
    assert "EOFError" in str(e.value)


def test_sampler_at_size_equals_the_reference(tmp_path, monkeypatch):
    """150,000 FASTQ records of 300 bp with -q 20 -d (BASELINE configs[4] shape; tests/golden/c5_at_size.py): counters, sample and
    count_bases against what the REFERENCE's own process_seqfile produced for the same file (tests/golden/c5_at_size.json) - with
    the default geometry (many pieces per region) and with small regions."""
    import hashlib
    import json
    import sys
    from microbecensus_amd import _native
    sys.path.insert(0, GOLD)
    import c5_at_size
    g = json.load(open(os.path.join(GOLD, "c5_at_size.json")))
    fq = str(tmp_path / "c5.fq")
    c5_at_size.write_fastq(fq)
    assert hashlib.md5(open(fq, "rb").read()).hexdigest() == g["file_md5"]
    for geom in ({}, {"MC_READER_REGION_BYTES": str(3 << 20), "MC_READER_PIECE_BYTES": str(64 << 10)}):
        for k, v in geom.items():
            monkeypatch.setenv(k, v)
        fa = str(tmp_path / "out.fa")
        reads, st = _native.sample_reads([fq], 300, 10_000_000, True, g["quality_offset"], 20, -5, 100, True, fa)
        assert {k: st[k] for k in ("too_short", "low_qual", "dups", "sampled")} == g["counters"]
        assert hashlib.md5(open(fa, "rb").read()).hexdigest() == g["reads_md5"]
        assert st["exhausted"] == 1 and st["bases"] == g["count_bases"]
        for k in geom:
            monkeypatch.delenv(k)
    assert _native.count_bases([fq]) == g["count_bases"]


def test_codec_follows_the_file_name(tmp_path):
    """open_file (reference :47-59) picks gzip / bz2 / plain by the extension: a plain-text file called *.gz makes gzip.open raise
    BadGzipFile (run_pipeline prints it and returns None) - the native reader reports the same instead of reading it."""
    import gzip
    from microbecensus_amd import _native
    text = b"".join(b">r%d\n%s\n" % (i, b"ACGT" * 30) for i in range(50))
    p = str(tmp_path / "plain.fa.gz")
    open(p, "wb").write(text)
    with pytest.raises(_native.ReferenceError_) as e:
        _native.count_bases([p])
    assert "BadGzipFile" in str(e.value)
    with pytest.raises(Exception):
        gzip.open(p).read()
    q = str(tmp_path / "real.fa.gz")
    with gzip.open(q, "wb") as f:
        f.write(text)
    assert _native.count_bases([q]) == 50 * 120
    e0 = str(tmp_path / "empty.fa.gz")
    open(e0, "wb").close()
    assert gzip.open(e0).read() == b"" and _native.count_bases([e0]) == 0


def _fastq_text(n, seed):
    import random
    rnd = random.Random(seed)
    out = []
    for i in range(n):
        L = rnd.choice([80, 100, 100, 100, 151])
        out.append("@r%d x\n%s\n+\n%s\n" % (i, "".join(rnd.choice("ACGTN") for _ in range(L)), "".join(rnd.choice("FGHIJ5?@") for _ in range(L))))
    return "".join(out).encode()


@pytest.mark.parametrize("chunk", ["4096", "20000", "1048576"])
def test_parallel_gzip_equals_gzip_module(tmp_path, monkeypatch, chunk):
    """A regular .gz file is inflated by several workers that guess deflate block starts (csrc/mc_pgzip.h); whatever the file looks
    like the bytes must be gzip.open's (reference open_file :47-59): one member at levels 1 / 6 / 9, several members (cat a.gz b.gz),
    bgzip-like 64 KB members, sync / full flushes (stored blocks), a member of incompressible binary in front of the text, an
    empty member in the middle.  Checked through the sampler (reads and counters) against the serial reader and Python's bases."""
    import gzip
    import zlib
    from microbecensus_amd import _native
    monkeypatch.setenv("MC_READER_GZ_CHUNK", chunk)
    text = _fastq_text(6000, 7)
    want_bases = sum(len(l) - 1 for i, l in enumerate(text.split(b"\n")[:-1]) if i % 4 == 1) + sum(1 for i, l in enumerate(text.split(b"\n")[:-1]) if i % 4 == 1)
    files = {}
    for lvl in (1, 6, 9):
        files["l%d" % lvl] = gzip.compress(text, lvl)
    cut = text.index(b"@r3000 ")
    files["cat"] = gzip.compress(text[:cut], 6) + gzip.compress(text[cut:], 1)
    files["bgzf"] = b"".join(gzip.compress(text[i:i + 65280], 6) for i in range(0, len(text), 65280))
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    parts = []
    for k, i in enumerate(range(0, len(text), 50000)):
        parts.append(co.compress(text[i:i + 50000])); parts.append(co.flush(zlib.Z_FULL_FLUSH if k % 2 else zlib.Z_SYNC_FLUSH))
    parts.append(co.flush())
    files["flush"] = b"".join(parts)
    files["empty_member"] = gzip.compress(text[:cut], 6) + gzip.compress(b"") + gzip.compress(text[cut:], 6)
    files["stored"] = gzip.compress(text, 0)
    ref = None
    for name, blob in files.items():
        p = str(tmp_path / (name + ".fq.gz"))
        open(p, "wb").write(blob)
        assert gzip.open(p).read() == text
        got, st = _native.sample_reads([p], 100, 1000000, True, 32, -5, -5, 100, False)
        if ref is None:
            monkeypatch.setenv("MC_READER_SERIAL_GZ", "1")
            ref = _native.sample_reads([p], 100, 1000000, True, 32, -5, -5, 100, False)
            monkeypatch.delenv("MC_READER_SERIAL_GZ")
        assert st == ref[1] and (got == ref[0]).all(), name
        assert _native.count_bases([p]) == st["bases"], name
    # damage: a cut file and a flipped byte deliver an error, not a short sample
    for name, blob in (("cut", files["l6"][: len(files["l6"]) * 2 // 3]), ("flip", files["l6"][:50000] + bytes([files["l6"][50000] ^ 0x41]) + files["l6"][50001:])):
        p = str(tmp_path / (name + ".fq.gz"))
        open(p, "wb").write(blob)
        with pytest.raises(_native.ReferenceError_):
            _native.count_bases([p])
    # binary in front of the text (the speculative decoder declines it, the sequential path takes over): same bytes as gzip
    import os as _os
    blob = gzip.compress(_os.urandom(300000), 6) + gzip.compress(text, 6)
    p = str(tmp_path / "bin_then_text.fq.gz")
    open(p, "wb").write(blob)
    assert gzip.open(p).read()[300000:] == text
    assert _native.count_bases([p]) >= 0                             # (parsed as whatever records the binary happens to hold - the point is that it is read to the end without an error)
    del want_bases


@pytest.mark.parametrize("serial", [False, True], ids=["parallel_inflate", "one_stream"])
def test_gzip_padding_and_trailing_bytes_as_gzip_open(tmp_path, monkeypatch, serial):
    """What may lie between and behind gzip members (GzipFile._read_eof / _read_gzip_header, behind the reference's open_file
    :47-59): zero padding is skipped and the next member read; other bytes raise BadGzipFile once the reader gets there - so
    run_pipeline fails if the sampler needs them and does not if it stops earlier.  Both native codecs (zlib's own gzread stops at
    zero padding and ignores other bytes)."""
    import gzip
    from microbecensus_amd import _native
    monkeypatch.setenv("MC_READER_GZ_CHUNK", "20000")
    if serial:
        monkeypatch.setenv("MC_READER_SERIAL_GZ", "1")
    text = _fastq_text(3000, 11)
    cut = text.index(b"@r1500 ")
    a, b = gzip.compress(text[:cut], 6), gzip.compress(text[cut:], 6)
    want, st_want = None, None
    for name, blob in (("plain", a + b), ("zeros_between", a + b"\0" * 5 + b), ("zeros_behind", a + b + b"\0" * 700)):
        p = str(tmp_path / (name + ".fq.gz"))
        open(p, "wb").write(blob)
        assert gzip.open(p).read() == text
        got, st = _native.sample_reads([p], 100, 1000000, True, 32, -5, -5, 100, False)
        if want is None:
            want, st_want = got, st
        assert st == st_want and (got == want).all(), name
        assert st["exhausted"] == 1
    for name, blob in (("garbage_behind", a + b + b"tail"), ("garbage_behind_zeros", a + b + b"\0\0\0x"), ("garbage_between", a + b"junk" + b)):
        p = str(tmp_path / (name + ".fq.gz"))
        open(p, "wb").write(blob)
        with pytest.raises(gzip.BadGzipFile):
            gzip.open(p).read()
        with pytest.raises(_native.ReferenceError_):
            _native.count_bases([p])
        with pytest.raises(_native.ReferenceError_):
            _native.sample_reads([p], 100, 1000000, True, 32, -5, -5, 100, False)
        # a sampler that has its reads before it gets there never sees the damage (the reference's head-take)
        got, st = _native.sample_reads([p], 100, 200, True, 32, -5, -5, 100, False)
        assert st["sampled"] == 200 and (got == want[:200]).all(), name


def test_streaming_fetch_equals_run(tmp_path):
    """mc_reader_start / fetch / join hand out the same reads mc_reader_run collects."""
    import ctypes as C
    import numpy as np
    from microbecensus_amd import _native
    path = os.path.join(GOLD, "inputs", "metagenome.fa.gz")
    want, st = _native.sample_reads([path], 100, 1000000, False, 0, -5, -5, 100, False)
    rd = _native.Reader([path], 100, 1000000, False, 0, -5, -5, 100, False)
    lib = rd.lib
    assert lib.mc_reader_start(rd.r) == 0
    got, at = [], 0
    buf = np.empty((7000, 100), np.uint8)
    while True:
        n = lib.mc_reader_fetch(rd.r, at, 7000, buf.ctypes.data_as(C.c_void_p))
        assert n >= 0
        if n == 0:
            break
        got.append(buf[:n].copy()); at += n
    assert lib.mc_reader_join(rd.r) == st["sampled"] == at
    assert (np.concatenate(got) == want).all()
    rd.close()


@pytest.mark.parametrize("quals,expect", [
    (["IIIIIIII", "IIII5III"], 32),          # a low character decides
    (["IIIIIIII", "IIIIhIII"], 64),          # a high character decides
    (["IIIIIIII", "JJJJ::::"], 32),          # nothing decides: the whole file is walked, 32
    (["hIII5III"], 64),                      # the first one that decides wins
    (["IIII", "II5I"], 32),                  # (with the multi-line record below)
])
def test_quality_offset_native_equals_python(quals, expect, tmp_path, monkeypatch):
    """auto_detect_quality_offset (reference :175-187) by the native parser, against the Python statement of the same walk;
    plain, gz, multi-line records, many regions and pieces."""
    import gzip
    from microbecensus_amd import _native, microbe_census as mc
    recs = []
    for i, q in enumerate(quals * 40):
        seq = "ACGTNACG"[:len(q)]
        if i % 7 == 3 and len(q) == 8:       # a record whose sequence and quality span two lines
            recs.append("@r%d\n%s\n%s\n+\n%s\n%s\n" % (i, seq[:4], seq[4:], q[:4], q[4:]))
        else:
            recs.append("@r%d\n%s\n+\n%s\n" % (i, seq, q))
    text = "".join(recs)
    plain = tmp_path / "q.fq"; plain.write_text(text)
    gz = tmp_path / "q.fq.gz"
    with gzip.open(gz, "wt") as f:
        f.write(text)
    for path in (str(plain), str(gz)):
        for geom in ({}, {"MC_READER_REGION_BYTES": "97", "MC_READER_PIECE_BYTES": "13"}):
            for k, v in geom.items():
                monkeypatch.setenv(k, v)
            assert _native.quality_offset(path) == expect
            for k in geom:
                monkeypatch.delenv(k)
        monkeypatch.setenv("MCENSUS_PYTHON_READER", "1")
        assert mc.auto_detect_quality_offset(path) == expect
        monkeypatch.delenv("MCENSUS_PYTHON_READER")
        assert mc.auto_detect_quality_offset(path) == expect


def test_quality_offset_of_a_fasta_record_is_left_to_python(tmp_path):
    from microbecensus_amd import _native
    p = tmp_path / "x.fa"; p.write_text(">a\nACGT\n>b\nACGT\n")
    assert _native.quality_offset(str(p)) is None


def test_inflate_tables_and_byte_mode_against_zlib(tmp_path):
    """The block decoder of the parallel inflate (csrc/mc_pgzip.h, round 5: one-entry-per-code tables with a second level, two
    literals per lookup, plain-byte output once no marker can follow, CRC-32 by carry-less multiplication) against zlib on streams
    made to reach its corners - through tools/pgz_bench.cpp, which prints the CRC-32 and the length of what the reader delivers:
    text over alphabets of 4 .. 90 letters with skewed frequencies (codes of up to 15 bits: second-level tables), every zlib
    strategy (fixed-Huffman blocks, Huffman only, RLE: distances of 1 .. 7), levels 1 / 6 / 9, members concatenated inside a
    chunk, repetitive lines whose markers never die out and noisy ones where the output goes over to bytes, at chunk sizes of
    4 KB .. 1 MB with 1 .. 6 workers."""
    import gzip
    import random
    import subprocess
    import zlib
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "pgz_bench")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(repo, "tools", "pgz_bench.cpp"), "-lz", "-pthread"])
    rng = random.Random(20261003)
    texts = {}
    for nletters in (4, 20, 90):
        letters = [chr(33 + i) for i in range(nletters)]
        weights = [1.0 / (1 + i) ** 2.2 for i in range(nletters)]             # a few frequent letters, a long tail: long codes
        lines = []
        for i in range(9000):
            lines.append("".join(rng.choices(letters, weights, k=rng.randint(20, 160))))
            if i % 3 == 0:
                lines.append(lines[-1][: rng.randint(1, 20)] * rng.randint(1, 6))        # repeats at short distances
            if i % 7 == 0:
                lines.append("I" * 150)                                          # a line every copy of which is a copy of the one before
        texts["a%d" % nletters] = ("\n".join(lines) + "\n").encode()
    texts["fastq"] = _fastq_text(5000, 3)
    files = {}
    for name, text in texts.items():
        for lvl, strat, tag in ((1, zlib.Z_DEFAULT_STRATEGY, "l1"), (6, zlib.Z_DEFAULT_STRATEGY, "l6"), (9, zlib.Z_DEFAULT_STRATEGY, "l9"), (6, zlib.Z_FIXED, "fixed"),
                                (6, zlib.Z_HUFFMAN_ONLY, "huff"), (6, zlib.Z_RLE, "rle"), (6, zlib.Z_FILTERED, "filt")):
            co = zlib.compressobj(lvl, zlib.DEFLATED, 31, 8, strat)
            files["%s_%s" % (name, tag)] = (co.compress(text) + co.flush(), text)
    both = texts["a20"] + texts["fastq"]                                         # members that begin and end inside the chunks, at changing levels
    files["members"] = (b"".join(gzip.compress(both[i:i + 70000], 1 + (k % 9)) for k, i in enumerate(range(0, len(both), 70000))), both)
    for name, (blob, text) in files.items():
        p = str(tmp_path / (name + ".gz"))
        open(p, "wb").write(blob)
        want = "crc32 %08x" % zlib.crc32(text)
        for threads, chunk in ((1, 4096), (3, 20000), (6, 1 << 20)):
            out = subprocess.run([exe, p, str(threads), str(chunk), "1"], stdout=subprocess.PIPE, check=True).stdout.decode()
            assert (" %d bytes " % len(text)) in out and want in out and "BAD" not in out, (name, threads, chunk, out)


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_parallel_verdicts_equal_the_record_by_record_sampler(seed, tmp_path, monkeypatch):
    """Without -d the parser's threads give the quality filter's verdicts and the sampler only sums the pieces up (round 5); with
    MC_READER_SERIAL_SAMPLER it walks the records one by one in file order as before (the form the sampler goldens pinned on the
    reference's own process_seqfile).  Random FASTQ text - short reads, N runs, low qualities, a sequence over two lines, CR LF -
    through both, at several head-takes (the sample getting full inside a piece, at a piece's end, never) and piece sizes:
    the same reads, the same counters."""
    import random
    import numpy as np
    from microbecensus_amd import _native
    rng = random.Random(seed)
    L = 60
    recs = []
    for i in range(6000):
        n = rng.choice([L - 7, L, L, L, L + 5, L + 40])
        seq = "".join(rng.choice("ACGT") for _ in range(n))
        if rng.random() < 0.15:
            k = rng.randrange(1, n // 2)
            at = rng.randrange(0, n - k)
            seq = seq[:at] + "N" * k + seq[at + k:]
        lo = rng.choice([2, 2, 20, 30])
        qual = "".join(chr(33 + rng.randrange(lo, 41)) for _ in range(n))
        nl = "\r\n" if seed == 3 else "\n"
        if seed == 4 and i % 97 == 0:
            recs.append("@r%d%s%s%s%s%s+%s%s%s" % (i, nl, seq[:n // 2], nl, seq[n // 2:], nl, nl, qual, nl))     # a sequence over two lines
        else:
            recs.append("@r%d%s%s%s+%s%s%s" % (i, nl, seq, nl, nl, qual, nl))
    path = tmp_path / "r.fq"
    path.write_bytes("".join(recs).encode())
    for piece, region in ((1 << 12, 1 << 16), (1 << 18, 1 << 22)):
        monkeypatch.setenv("MC_READER_PIECE_BYTES", str(piece))
        monkeypatch.setenv("MC_READER_REGION_BYTES", str(region))
        for nreads in (1, 37, 1000, 2500, 100000):
            for qargs in ((33, -5, -5, 100), (33, 10, 25, 5)):
                monkeypatch.delenv("MC_READER_SERIAL_SAMPLER", raising=False)
                a, sa = _native.sample_reads([str(path)], L, nreads, True, qargs[0], qargs[1], qargs[2], qargs[3], False)
                monkeypatch.setenv("MC_READER_SERIAL_SAMPLER", "1")
                b, sb = _native.sample_reads([str(path)], L, nreads, True, qargs[0], qargs[1], qargs[2], qargs[3], False)
                assert sa == sb, (piece, nreads, qargs, sa, sb)
                assert a.shape == b.shape and (a == b).all()
                assert sa["sampled"] == min(nreads, sa["sampled"]) and (nreads >= 100000 or sa["sampled"] <= nreads)
    assert sa["too_short"] > 0 and sa["low_qual"] > 0


def _rc(s):
    return s[::-1].translate(str.maketrans("ACGTN", "TGCAN"))


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_class_sharded_dup_verdicts_equal_the_record_by_record_sampler(seed, tmp_path, monkeypatch):
    """-d (reference :345 duplicate test before QC, :354 only accepted reads enter the set): the verdicts are given per duplicate
    class {s, rc(s)} by NSHARD walkers (round 6); MC_READER_SERIAL_SAMPLER keeps the record-by-record walk the goldens were pinned
    on, and a literal Python loop of process_seqfile is the third voice.  The file is made to hit the rule's corners: exact and
    reverse-complement repeats whose FIRST occurrence failed the quality filter (so a later one is accepted, and the ones behind
    it are duplicates), repeats of different untrimmed length (not duplicates), palindromes, repeats across two files, the
    head-take ending inside a class, and records the reference raises at (a base outside ACGTN) in front of and behind the take."""
    import random
    from microbecensus_amd import _native
    rng = random.Random(seed)
    L = 50
    pool_seqs = []
    recs = []
    for i in range(9000):
        u = rng.random()
        if pool_seqs and u < 0.35:
            seq = rng.choice(pool_seqs)
            if rng.random() < 0.5:
                seq = _rc(seq)
            if rng.random() < 0.1:
                seq = seq + "A"                                  # another untrimmed sequence: not a duplicate
        elif u < 0.38:
            h = "".join(rng.choice("ACGT") for _ in range(L // 2 + 3))
            seq = h + _rc(h)                                     # its own reverse complement
        else:
            n = rng.choice([L - 5, L, L, L + 3, L + 30, L + rng.randrange(0, 130)])   # (every remainder of the hashes' 32- / 16- / 8-byte steps)
            seq = "".join(rng.choice("ACGT") for _ in range(n))
            if rng.random() < 0.1:
                at = rng.randrange(0, n - 8)
                seq = seq[:at] + "N" * 8 + seq[at + 8:]
        pool_seqs.append(seq)
        lo = rng.choice([2, 2, 25, 30])                          # half of the records fail -q 20 / -m 25
        qual = "".join(chr(33 + rng.randrange(lo, 41)) for _ in range(len(seq)))
        recs.append((seq, qual))

    def write(path, rr):
        with open(path, "w") as f:
            for i, (s, q) in enumerate(rr):
                f.write("@r%d\n%s\n+\n%s\n" % (i, s, q))

    def py_sampler(files, nreads, qargs, bad_at=None):
        # process_seqfile :336-356 stated literally over the record lists
        seqs, kept, st = set(), [], dict(too_short=0, low_qual=0, dups=0)
        for rr in files:
            for (s, q) in rr:
                if len(s) < L:
                    st["too_short"] += 1
                    continue
                if s in seqs:
                    st["dups"] += 1
                    continue
                if set(s) - set("ACGTN"):
                    return None, None
                if _rc(s) in seqs:
                    st["dups"] += 1
                    continue
                t, ph = s[:L], [ord(c) - 33 for c in q[:L]]
                if 100 * t.count("N") / float(L) > qargs[2] or sum(ph) / float(len(ph)) < qargs[1] or min(ph) < qargs[0]:
                    st["low_qual"] += 1
                    continue
                kept.append(t)
                seqs.add(s)
                if len(kept) == nreads:
                    st["sampled"] = len(kept)
                    return kept, st
        st["sampled"] = len(kept)
        return kept, st

    fa, fb = str(tmp_path / "a.fq"), str(tmp_path / "b.fq")
    write(fa, recs[:6000])
    write(fb, recs[6000:])
    for piece, region in ((1 << 11, 1 << 15), (1 << 18, 1 << 22)):
        monkeypatch.setenv("MC_READER_PIECE_BYTES", str(piece))
        monkeypatch.setenv("MC_READER_REGION_BYTES", str(region))
        for nreads in (1, 50, 777, 2000, 100000):
            for qargs in ((-5, -5, 100), (20, 25, 10)):
                want, wst = py_sampler([recs[:6000], recs[6000:]], nreads, qargs)
                res = []
                for ser in (False, True):
                    if ser:
                        monkeypatch.setenv("MC_READER_SERIAL_SAMPLER", "1")
                    else:
                        monkeypatch.delenv("MC_READER_SERIAL_SAMPLER", raising=False)
                    res.append(_native.sample_reads([fa, fb], L, nreads, True, 33, qargs[0], qargs[1], qargs[2], True))
                (a, sa), (b, sb) = res
                assert sa == sb, (piece, nreads, qargs, sa, sb)
                assert a.shape == b.shape and (a == b).all()
                assert [bytes(x).decode() for x in a] == want
                assert {k: sa[k] for k in wst} == wst
        monkeypatch.delenv("MC_READER_SERIAL_SAMPLER", raising=False)
    assert wst["dups"] > 500 and wst["low_qual"] > 500 and wst["too_short"] > 0
    # a record the reference raises at: behind the take nothing happens, in front of it both forms raise
    bad = list(recs[:6000])
    bad[3000] = ("ACGTX" * 12, "I" * 60)
    write(fa, bad)
    want, wst = py_sampler([bad], 300, (20, 25, 10))
    assert want is not None and len(want) == 300
    for ser in (False, True):
        if ser:
            monkeypatch.setenv("MC_READER_SERIAL_SAMPLER", "1")
        else:
            monkeypatch.delenv("MC_READER_SERIAL_SAMPLER", raising=False)
        a, sa = _native.sample_reads([fa], L, 300, True, 33, 20, 25, 10, True)
        assert [bytes(x).decode() for x in a] == want and {k: sa[k] for k in wst} == wst
        assert py_sampler([bad], 100000, (20, 25, 10))[0] is None
        with pytest.raises(_native.ReferenceError_) as e:
            _native.sample_reads([fa], L, 100000, True, 33, 20, 25, 10, True)
        assert "KeyError" in str(e.value)


def test_parallel_bz2_equals_the_one_stream_decoder(tmp_path, monkeypatch):
    """.bz2 files are decoded block by block on several threads (csrc/mc_pbzip2.h, round 6): the blocks of a stream are independent, each is
    cut out at its bit offset and decoded as a one-block stream of its own; whatever is not a well-formed stream is left to the one-stream
    decoder (MC_READER_SERIAL_BZ2 forces it for everything).  Both forms, and Python's bz2 module, on: many small blocks (level 1), several
    streams, an empty stream in between, trailing bytes that are no stream, a file cut inside a block / between two blocks / inside the
    end marker, a damaged byte in the middle of a block (the block's CRC fails: back to the stream's start with the one-stream decoder),
    a damaged block CRC field (the stream no longer checks out: one-stream decoder from there)."""
    import bz2
    import random
    from microbecensus_amd import _native
    rng = random.Random(99)
    recs = []
    for i in range(12000):
        n = rng.choice([60, 75, 75, 100])
        recs.append("@r%d\n%s\n+\n%s\n" % (i, "".join(rng.choice("ACGT") for _ in range(n)), "".join(chr(33 + rng.randrange(20, 41)) for _ in range(n))))
    text = "".join(recs).encode()
    third = len(text) // 3
    cut = [text[:third].rfind(b"\n@r") + 1, text[:2 * third].rfind(b"\n@r") + 1]
    a, b, c = text[:cut[0]], text[cut[0]:cut[1]], text[cut[1]:]
    one = bz2.compress(text, 1)                                     # ~ 30 blocks of 100 k
    multi = bz2.compress(a, 1) + bz2.compress(b"") + bz2.compress(b, 9) + bz2.compress(c, 2)
    blobs = {
        "one": one, "multi": multi, "trailing": multi + b"this is no stream", "zero_padding": multi + b"\0" * 100,
        "cut_in_block": one[: len(one) * 2 // 3], "cut_in_end_marker": one[:-6], "cut_multi": multi[:-20],
    }
    dmg = bytearray(one); dmg[len(one) // 2] ^= 0x55
    blobs["damaged_payload"] = bytes(dmg)
    dmg = bytearray(multi); dmg[len(bz2.compress(a, 1)) + 14 + 4 + 7] ^= 0xFF   # inside the first block header of the third stream (its CRC field)
    blobs["damaged_crc_field"] = bytes(dmg)
    monkeypatch.setenv("MC_READER_THREADS", "4")
    monkeypatch.setenv("MC_READER_REGION_BYTES", str(1 << 16))
    for name, blob in blobs.items():
        p = str(tmp_path / (name + ".fq.bz2"))
        open(p, "wb").write(blob)
        try:
            want = bz2.open(p).read()
            py_err = None
        except Exception as e:                                     # noqa: BLE001
            want, py_err = None, type(e).__name__
        res = []
        for serial in (False, True):
            if serial:
                monkeypatch.setenv("MC_READER_SERIAL_BZ2", "1")
            else:
                monkeypatch.delenv("MC_READER_SERIAL_BZ2", raising=False)
            try:
                reads, st = _native.sample_reads([p], 60, 10**9, True, 33, -5, -5, 100, False)
                res.append(("ok", st["records"], st["sampled"], st["bases"], reads.tobytes()))
            except _native.ReferenceError_ as e:
                res.append(("raises", str(e).split(":")[0]))
        assert res[0] == res[1], (name, res[0][:4], res[1][:4])
        if py_err is None:
            assert res[0][0] == "ok" and res[0][1] == want.count(b"\n+\n"), (name, res[0][:4])
        else:
            assert res[0][0] == "raises", (name, py_err, res[0][:4])
