"""CPU checks of the code the GPU runs: the per-thread device functions (microbecensus_amd/csrc/mc_core.h,
mc_finish.h) and the host index builder (mc_index.h) are compiled with g++ into tests/emul/mc_emul and run
stage by stage.  The emulation is test infrastructure; the product only ever executes the HIP build."""
import gzip
import hashlib
import json
import os
import struct
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
GOLD = os.path.join(HERE, "golden")


@pytest.fixture(scope="session")
def emul_bin(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("emul") / "mc_emul")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe, os.path.join(HERE, "emul", "mc_emul.cpp")])
    return exe


@pytest.fixture(scope="session")
def markers_faa(tmp_path_factory):
    p = str(tmp_path_factory.mktemp("db") / "markers.faa")
    with gzip.open(os.path.join(REPO, "microbecensus_amd", "data", "markers.faa.gz"), "rb") as f, open(p, "wb") as o:
        o.write(f.read())
    return p


def test_device_code_reproduces_reference_m8(emul_bin, markers_faa, tmp_path):
    """config 1 (example.fq.gz -n 10000 -l 100): same m8 bytes as the reference's RAPsearch2 run."""
    meta = json.load(open(os.path.join(GOLD, "config1_example_fq.json")))
    fa = tmp_path / "reads.fa"
    fa.write_bytes(gzip.open(os.path.join(GOLD, "config1_example_fq.reads.fa.gz"), "rb").read())
    out = tmp_path / "out.m8"
    subprocess.check_call([emul_bin, markers_faa, str(fa), str(out)], stderr=subprocess.DEVNULL)
    assert hashlib.md5(out.read_bytes()).hexdigest() == meta["m8_md5"]


def test_device_code_on_dirty_reads(emul_bin, markers_faa, tmp_path):
    """Lower case, IUPAC codes, `*`, `-`, digits, blanks in the reads (golden from the reference's binary,
    tests/golden/make_dirty_golden.py): the per-thread code the kernels share gives the reference's m8."""
    meta = json.load(open(os.path.join(GOLD, "dirty_reads.json")))
    fa = tmp_path / "dirty.fa"
    fa.write_bytes(gzip.open(os.path.join(GOLD, "dirty_reads.fa.gz"), "rb").read())
    out = tmp_path / "out.m8"
    subprocess.check_call([emul_bin, markers_faa, str(fa), str(out)], stderr=subprocess.DEVNULL)
    assert hashlib.md5(out.read_bytes()).hexdigest() == meta["m8_md5"]


def test_device_code_on_a_database_with_a_seed_threshold(emul_bin, tmp_path):
    """The generic seed enumeration (mc_enumerate_seeds: seeds of 6 .. 9 residues) and the evaluation of short seeds on the second
    database (`.info` threshold 1, tests/golden/make_generic_db_golden.py): the reference's m8."""
    import sys
    sys.path.insert(0, GOLD)
    import make_generic_db_golden as G
    meta = json.load(open(os.path.join(GOLD, "generic_db.json")))
    names, seqs, reads = G.case_inputs()
    (tmp_path / "db2.faa").write_bytes(G.fasta_bytes(names, seqs))
    (tmp_path / "reads.fa").write_bytes(b"".join(b">%d\n%s\n" % (i, bytes(r)) for i, r in enumerate(reads)))
    out = tmp_path / "out.m8"
    subprocess.check_call([emul_bin, str(tmp_path / "db2.faa"), str(tmp_path / "reads.fa"), str(out)], stderr=subprocess.DEVNULL)
    assert hashlib.md5(out.read_bytes()).hexdigest() == meta["m8_md5"]


def test_index_builder_matches_prerapsearch(emul_bin, markers_faa, ref_dir, tmp_path):
    """Bucket starts, posting order, suffix keys, residue codes, frequency threshold and letter frequencies of
    the product's index builder against the database prerapsearch wrote (oracle/_ref/rapdb_2.15[.info])."""
    fa = tmp_path / "one.fa"
    fa.write_text(">0\n" + "ACGT" * 25 + "\n")
    dump = tmp_path / "index.bin"
    env = dict(os.environ, MC_DUMP_INDEX=str(dump))
    subprocess.check_call([emul_bin, markers_faa, str(fa), str(tmp_path / "o.m8")], env=env, stderr=subprocess.DEVNULL)
    b = dump.read_bytes()
    npost, nres = struct.unpack_from("<QQ", b, 0)
    p = 16
    bstart = np.frombuffer(b, "<u4", 1000001, p); p += 4 * 1000001
    post = np.frombuffer(b, "<u4", npost, p); p += 4 * npost
    keys = np.frombuffer(b, "<u2", npost, p); p += 2 * npost
    codes = np.frombuffer(b, "u1", nres, p); p += nres
    thr = struct.unpack_from("<I", b, p)[0]; p += 4
    letter_p = np.frombuffer(b, "<f8", 10, p)
    # parse the boost archive prerapsearch wrote
    r = open(os.path.join(ref_dir, "rapdb_2.15"), "rb").read()
    q = 0x28
    n = struct.unpack_from("<Q", r, q)[0]; q += 8
    ref_codes = np.frombuffer(r, "u1", n, q); q += n
    m = struct.unpack_from("<Q", r, q)[0]; q += 8 + 4 * m
    q += 5 + 8 + 4
    ref_start = np.zeros(1000001, dtype=np.int64)
    chunks = []
    for i in range(1000000):
        c = struct.unpack_from("<Q", r, q)[0]; q += 8
        if c:
            chunks.append(np.frombuffer(r, "<u4", c, q))
        q += 4 * c
        ref_start[i + 1] = ref_start[i] + c
    ref_post = np.concatenate(chunks)
    q += 5
    nn = struct.unpack_from("<Q", r, q)[0]; q += 8 + 4
    for _ in range(nn):
        l = struct.unpack_from("<Q", r, q)[0]; q += 8 + l
    q += 5 + 8 + 4
    kch = []
    for i in range(1000000):
        c = struct.unpack_from("<Q", r, q)[0]; q += 8
        if c:
            kch.append(np.frombuffer(r, "<u2", c, q))
        q += 2 * c
    ref_keys = np.concatenate(kch)
    assert nres == n and np.array_equal(codes, ref_codes)
    assert np.array_equal(bstart.astype(np.int64), ref_start)
    assert np.array_equal(post, ref_post)
    assert np.array_equal(keys, ref_keys)
    info = open(os.path.join(ref_dir, "rapdb_2.15.info"), "rb").read()
    assert thr == struct.unpack_from("<I", info, 68 + 4000000)[0]
    ref_p = np.frombuffer(info, "<f8", 10, 68 + 4000000 + 4 + 8)
    assert np.allclose(letter_p, ref_p, rtol=0, atol=1e-6)


def test_gpu_index_structures_agree_with_the_binary_searches(emul_bin, markers_faa, tmp_path):
    """Exhaustive check, over every key of the marker index and its near misses (25 M probes), of the structures the
    seed kernel uses instead of the reference's per-bucket binary searches (ExtendSeq2Set):
      * bucket records + group scan (mc_key_range_rec): same posting range, same start index, same reference key-probe count;
      * 10-mer / 9-mer Bloom filters and the wildcard filter: no false negative (a probe with a range always passes);
      * range table of the long groups: same range and start index as the binary searches;
      * pair filter: no false negative; probe form of the short-group scan (mc_group_match8): same range as the general form."""
    fa = tmp_path / "one.fa"
    fa.write_text(">0\n" + "ACGT" * 25 + "\n")
    env = dict(os.environ, MC_CHECK_SCAN="1")
    r = subprocess.run([emul_bin, markers_faa, str(fa), str(tmp_path / "o.m8")], env=env, stderr=subprocess.PIPE)
    err = r.stderr.decode()
    assert r.returncode == 0, err
    assert " 0 mismatches" in err
    assert " 0 false negatives" in err
    assert " 0 differ from the binary searches" in err
    assert " 0 differ from the general form" in err


def test_seg_variants_agree_frame_by_frame(emul_bin, markers_faa, tmp_path):
    """The fixed-point / register SEG the kernel runs against the plain restatement, on every frame of the config-1 reads;
    the emulation also proves the integer entropy tests over every possible window composition (mc_seg_fx_verify)."""
    import gzip
    fa = tmp_path / "c1.fa"
    fa.write_bytes(gzip.open(os.path.join(GOLD, "config1_example_fq.reads.fa.gz")).read())
    env = dict(os.environ, MC_CHECK_SEG="1")
    r = subprocess.run([emul_bin, markers_faa, str(fa), str(tmp_path / "o.m8")], env=env, stderr=subprocess.PIPE)
    err = r.stderr.decode()
    assert r.returncode == 0, err
    assert "0 disagreements over all window compositions" in err
    assert ", 0 differ from the plain restatement" in err
    assert "base codes: 0 of 512 differ" in err


def test_shared_algorithms_and_reader_under_sanitizers(markers_faa, tmp_path):
    """AddressSanitizer + UBSan (CPU build; GPU sanitizers are not available on the pool): the per-thread algorithms the
    kernels share with this emulation, on 1,500 config-1 reads with the SEG cross-check, and the native reader on truncated
    copies of the sampler inputs."""
    exe = str(tmp_path / "mc_emul_asan")
    flags = ["-O1", "-g", "-std=c++17", "-ffp-contract=off", "-fsanitize=address,undefined", "-fno-omit-frame-pointer"]
    subprocess.check_call(["g++"] + flags + ["-o", exe, os.path.join(HERE, "emul", "mc_emul.cpp")])
    fa = tmp_path / "c1.fa"
    lines = gzip.open(os.path.join(GOLD, "config1_example_fq.reads.fa.gz"), "rt").readlines()[:3000]
    fa.write_text("".join(lines))
    env = dict(os.environ, MC_CHECK_SEG="1")
    r = subprocess.run([exe, markers_faa, str(fa), str(tmp_path / "o.m8")], env=env, stderr=subprocess.PIPE)
    err = r.stderr.decode()
    assert r.returncode == 0 and "AddressSanitizer" not in err and "runtime error" not in err, err[-2000:]
    # reader: every sampler input truncated at ~100 places, four parameter sets each
    drv = tmp_path / "rd.cpp"
    drv.write_text(r'''
#include <cstdio>
#include <vector>
#include "%s"
int main(int argc, char **argv) {
    for (int a = 2; a < argc; a++) {
        FILE *f = fopen(argv[a], "rb"); if (!f) return 2;
        std::vector<char> b; fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET); b.resize(n); if (n && fread(b.data(), 1, n, f) != (size_t)n) return 2; fclose(f);
        for (size_t cut = 0; cut <= b.size(); cut += b.size() / 97 + 1) {
            FILE *o = fopen(argv[1], "wb"); fwrite(b.data(), 1, cut, o); fclose(o);
            for (int mode = 0; mode < 4; mode++) {
                const char *paths[1] = {argv[1]};
                mc_reader *r = mc_reader_open(paths, 1, mode & 1 ? 50 : 100, 1000000, mode >> 1, 32, mode >> 1 ? 20 : -5, -5, mode & 1 ? 5 : 100, mode & 1, nullptr);
                if (!r) return 3;
                long long k = mc_reader_run(r);
                if (k > 0) { const uint8_t *p = mc_reader_reads(r); volatile unsigned s = 0; for (long long i = 0; i < k * (mode & 1 ? 50 : 100); i++) s += p[i]; }
                mc_reader_close(r);
                (void)mc_count_bases(paths, 1);
            }
        }
    }
    return 0;
}
''' % os.path.join(REPO, "include", "mcensus.h"))
    rexe = str(tmp_path / "rd_asan")
    subprocess.check_call(["g++"] + flags + ["-pthread", "-o", rexe, str(drv), os.path.join(REPO, "microbecensus_amd", "csrc", "mc_reader.cpp"), "-lz"])
    inputs = sorted(os.path.join(GOLD, "sampler", f) for f in os.listdir(os.path.join(GOLD, "sampler")))
    r = subprocess.run([rexe, str(tmp_path / "cut.bin")] + inputs, stderr=subprocess.PIPE)
    err = r.stderr.decode()
    assert r.returncode == 0 and "AddressSanitizer" not in err and "runtime error" not in err, err[-2000:]


def test_heap_words_replay_equals_libstdcxx(tmp_path):
    """MergeRes' heap sort as a lane of k_heap_lanes runs it (csrc/mc_heap_words.h, the text the kernel compiles) == the plain loop of
    mc_sort_impl.h == libstdc++'s make_heap + sort_heap, word for word, on 200,000 arrays full of ties (MergeRes@0x40e3b0)."""
    exe = str(tmp_path / "heap_words_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(REPO, "microbecensus_amd", "csrc"), "-o", exe, os.path.join(HERE, "emul", "heap_words_check.cpp")])
    out = subprocess.run([exe], stdout=subprocess.PIPE, check=True).stdout.decode()
    assert "arrays 200000 differ_from_plain 0 differ_from_libstdcxx 0" in out, out
