"""world_size-2 gloo test of the multi-GPU layer: contiguous sharding of the accepted-read stream and the exact
integer reduction of the per-family accumulators.  The sharded result must equal the unsharded one, and the
aggregates must equal the reference's aggregate_hits() on the golden best hits (ints identical, 'cov' sums to
1e-12 relative)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, "tests", "golden")

WORKER = r'''
import json, os, sys
import numpy as np
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from microbecensus_amd import distributed as D
from microbecensus_amd import _native, microbe_census as mc
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
g = json.load(open(os.path.join(sys.argv[1], "tests", "golden", "unittest_metagenome.json")))
model = _native.load_model(); fams = model["families"]
names, seqs = _native.load_markers()
items = sorted(g["best_hits"].items(), key=lambda kv: int(kv[0]))
reads = np.array([int(k) for k, _ in items])
lo, hi = D.shard_bounds(g["sampled_reads"], rank, world)          # shard the READ stream, not the hit list
mine = [(k, v) for k, v in items if lo <= int(k) < hi]
best = np.zeros(len(mine), dtype=_native.BEST_DTYPE)
for i, (k, v) in enumerate(mine):
    best[i] = (int(k), fams.index(v[0]), int(v[1]), int(round(v[1] / v[2])), v[3])
acc = D.family_accumulators(best, len(fams))
acc = D.all_reduce_accumulators(*acc)
agg = D.aggregate_from_accumulators(*acc, fams, mc.find_opt_pars(None, 100))
if rank == 0:
    json.dump({"agg": agg, "n": int(acc[0].sum())}, open(sys.argv[2], "w"))
dist.destroy_process_group()
'''


def test_two_rank_reduce_matches_reference_aggregates(tmp_path):
    worker = tmp_path / "worker.py"
    worker.write_text(WORKER)
    out = tmp_path / "out.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                           "--master-port", "29517", str(worker), REPO, str(out)], env=env, timeout=600)
    res = json.load(open(out))
    g = json.load(open(os.path.join(GOLD, "unittest_metagenome.json")))
    assert res["n"] == len(g["best_hits"])
    assert set(res["agg"]) == set(g["agg_hits"])
    for fam, want in g["agg_hits"].items():
        got = res["agg"][fam]
        if float(want).is_integer():
            assert got == want, fam
        else:
            assert abs(got - want) <= 1e-12 * abs(want), fam


def test_shard_bounds_cover_everything():
    from microbecensus_amd.distributed import shard_bounds
    for n in (0, 1, 7, 70623, 10**8):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


DEAL_WORKER = r'''
import json, os, sys, zlib
import numpy as np
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from microbecensus_amd import distributed as D
from microbecensus_amd import _native
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
path = os.path.join(sys.argv[1], "tests", "golden", "inputs", "metagenome.fa.gz")
got = []
def on_batch(block, first):
    a = block if isinstance(block, np.ndarray) else block.numpy().reshape(-1, 100)
    got.append((int(first), int(a.shape[0]), zlib.crc32(a.tobytes())))
rd = _native.Reader([path], 100, 1000000, False, 0, -5, -5, 100, False) if rank == 0 else None
n_total, trace, err = D.stream_batches(rd, 100, on_batch)
if rd is not None:
    rd.close()
all_got = [None] * world
dist.all_gather_object(all_got, got)
if rank == 0:
    json.dump({"n_total": n_total, "err": None if err is None else str(err), "batches": all_got, "deals": [t[4] for t in trace if t[0] == "deal"]}, open(sys.argv[2], "w"))
dist.barrier()
dist.destroy_process_group()
'''


def test_batches_are_dealt_to_free_ranks(tmp_path):
    """stream_batches with three ranks (gloo, no GPU): rank 0 samples the reference's unit-test metagenome and deals 2,000-read
    batches to whoever holds a credit; rank 1 dawdles 60 ms after every batch.  Every read arrives exactly once with its global index
    (the batches tile [0, sampled) and carry the bytes the plain sampler returns), and the slow rank is simply dealt fewer batches
    than the others instead of holding up the stream (round 3 dealt batch k to rank k mod N)."""
    import zlib
    from microbecensus_amd import _native
    worker = tmp_path / "deal.py"
    worker.write_text(DEAL_WORKER)
    out = tmp_path / "deal.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MC_DIST_BATCH="2000", MC_DIST_SLOW="1:60")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
                           "--master-port", "29519", str(worker), REPO, str(out)], env=env, timeout=600)
    res = json.load(open(out))
    g = json.load(open(os.path.join(GOLD, "unittest_metagenome.json")))
    assert res["err"] is None and res["n_total"] == g["sampled_reads"]
    want, st = _native.sample_reads([os.path.join(GOLD, "inputs", "metagenome.fa.gz")], 100, 1000000, False, 0, -5, -5, 100, False)
    pieces = sorted(tuple(b) for per_rank in res["batches"] for b in per_rank)
    at = 0
    for first, n, crc in pieces:                                  # the batches tile the sampled reads, in order, bytes intact
        assert first == at and crc == zlib.crc32(want[first:first + n].tobytes())
        at += n
    assert at == g["sampled_reads"] == len(want)
    per_rank = [len(b) for b in res["batches"]]
    assert sum(per_rank) == len(res["deals"]) == -(-g["sampled_reads"] // 2000)
    assert [res["deals"].count(r) for r in range(3)] == per_rank
    assert per_rank[1] < per_rank[0] and per_rank[1] < per_rank[2], per_rank      # the slow rank asked less often
    assert per_rank[1] >= 2                                       # ... but was not starved (it held two credits from the start)


SHARD_WORKER = r'''
import gzip, json, os, sys, zlib
import numpy as np
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from microbecensus_amd import distributed as D
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
cases = json.load(open(sys.argv[2]))
out = {}
for name, a in cases.items():
    got = []
    def on_batch(block, first):
        got.append((int(first), int(block.shape[0]), zlib.crc32(np.ascontiguousarray(block).tobytes())))
    assert D.sharded_sampling_usable(a)
    n_total, stats, bases, status = D.stream_batches_sharded(a, on_batch)
    allg = [None] * world
    dist.all_gather_object(allg, got)
    out[name] = {"n_total": n_total, "stats": stats, "bases": bases, "status": status, "batches": allg}
if rank == 0:
    json.dump(out, open(sys.argv[3], "w"))
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world", [2, 4])
def test_every_rank_samples_its_own_slices(tmp_path, world):
    """stream_batches_sharded (gloo, no GPU): the reference's inputs as plain files, cut into slices of 150 KB that the ranks sample
    side by side (mc_reader_open_range) - the kept reads, in the order of their global indices, are exactly the reads of the ONE
    sequential sampler (process_seqfile :328-367: head-take over the files one after the other), its counters and count_bases
    included: all reads of the unit-test metagenome; the first 5,000 accepted reads of example.fq (length filter; the take ends
    inside a slice); the paired library given as two files with the take ending in the second; a -q 20 run of the 300 bp library."""
    import gzip
    import zlib
    from microbecensus_amd import _native
    inp = os.path.join(GOLD, "inputs")
    plain = {}
    for f in ("metagenome.fa.gz", "example.fq.gz", "c4_pair_1.fq.gz", "c4_pair_2.fq.gz", "c5_300bp.fq.gz"):
        plain[f] = str(tmp_path / f[:-3])
        open(plain[f], "wb").write(gzip.open(os.path.join(inp, f), "rb").read())
    base = {"min_quality": -5, "mean_quality": -5, "max_unknown": 100, "filter_dups": False}
    cases = {
        "metagenome": dict(base, seqfiles=[plain["metagenome.fa.gz"]], read_length=100, nreads=10**9, file_type="fasta", quality_offset=None),
        "example_take": dict(base, seqfiles=[plain["example.fq.gz"]], read_length=100, nreads=5000, file_type="fastq", quality_offset=32),
        "pair_take": dict(base, seqfiles=[plain["c4_pair_1.fq.gz"], plain["c4_pair_2.fq.gz"]], read_length=150, nreads=15000, file_type="fastq", quality_offset=32),
        "q20": dict(base, seqfiles=[plain["c5_300bp.fq.gz"]], read_length=300, nreads=10**9, file_type="fastq", quality_offset=32, min_quality=20),
    }
    # .bz2: the blocks of a bzip2 file are independent, so its block ranges shard like the byte windows of a plain file (round 6): one block per
    # rank and round here (level 1: 100 KB of text per block), a file of three streams, the take ending in the second file of two
    import bz2
    ex = open(plain["example.fq.gz"], "rb").read()
    cut = ex.rfind(b"\n@", 0, len(ex) // 2) + 1
    bz = {"example.fq.bz2": bz2.compress(ex, 1), "example3.fq.bz2": bz2.compress(ex[:cut], 1) + bz2.compress(b"") + bz2.compress(ex[cut:], 2),
          "metagenome.fa.bz2": bz2.compress(open(plain["metagenome.fa.gz"], "rb").read(), 1)}
    for f, blob in bz.items():
        plain[f] = str(tmp_path / f)
        open(plain[f], "wb").write(blob)
    cases.update({
        "bz2_all": dict(base, seqfiles=[plain["example.fq.bz2"]], read_length=100, nreads=10**9, file_type="fastq", quality_offset=32),
        "bz2_streams_take": dict(base, seqfiles=[plain["example3.fq.bz2"]], read_length=100, nreads=6000, file_type="fastq", quality_offset=32),
        "bz2_and_plain": dict(base, seqfiles=[plain["metagenome.fa.bz2"], plain["metagenome.fa.gz"]], read_length=100, nreads=100000, file_type="fasta", quality_offset=None),
    })
    # .gz: a member cannot be entered in the middle, but it can be decoded from the middle speculatively - every rank decodes its slice of chunks
    # at once, the 32 KB windows are handed along the slices through the group's store, the members' CRCs follow (round 6).  Chunks of 64 KB
    # here, two per rank and round: the reference's own example (one member), the paired library with the take in the second file, a file
    # of three members, the 300 bp library with -q 20
    gz3 = str(tmp_path / "example3.fq.gz")
    open(gz3, "wb").write(gzip.compress(ex[:cut], 1) + gzip.compress(b"") + gzip.compress(ex[cut:], 9))
    cases.update({
        "gz_all": dict(base, seqfiles=[os.path.join(inp, "example.fq.gz")], read_length=100, nreads=10**9, file_type="fastq", quality_offset=32),
        "gz_pair_take": dict(base, seqfiles=[os.path.join(inp, "c4_pair_1.fq.gz"), os.path.join(inp, "c4_pair_2.fq.gz")], read_length=150, nreads=15000, file_type="fastq", quality_offset=32),
        "gz_members_take": dict(base, seqfiles=[gz3], read_length=100, nreads=6000, file_type="fastq", quality_offset=32),
        "gz_q20": dict(base, seqfiles=[os.path.join(inp, "c5_300bp.fq.gz")], read_length=300, nreads=10**9, file_type="fastq", quality_offset=32, min_quality=20),
        "gz_and_plain": dict(base, seqfiles=[os.path.join(inp, "metagenome.fa.gz"), plain["metagenome.fa.gz"]], read_length=100, nreads=100000, file_type="fasta", quality_offset=None),
    })
    cj = tmp_path / "cases.json"
    cj.write_text(json.dumps(cases))
    worker = tmp_path / "shard.py"
    worker.write_text(SHARD_WORKER)
    out = tmp_path / "shard.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MC_DIST_SLICE="150000", MC_DIST_GZ_CHUNK="65536")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                           "--master-port", str(29521 + world), str(worker), REPO, str(cj), str(out)], env=env, timeout=900)
    res = json.load(open(out))
    for name, a in cases.items():
        want, st = _native.sample_reads(a["seqfiles"], a["read_length"], a["nreads"], a["file_type"] == "fastq", a["quality_offset"] or 0,
                                        a["min_quality"], a["mean_quality"], a["max_unknown"], False)
        r = res[name]
        assert r["status"] == 0 and r["n_total"] == st["sampled"] == len(want), name
        at = 0
        for first, n, crc in sorted(tuple(b) for per_rank in r["batches"] for b in per_rank):
            assert first == at and crc == zlib.crc32(want[first:first + n].tobytes()), (name, first, at)
            at += n
        assert at == len(want), name
        for k in ("too_short", "low_qual", "records"):
            assert r["stats"][k] == st[k], (name, k, r["stats"], st)
        assert r["bases"] == (st["bases"] if st["exhausted"] else -1) or (r["bases"] == -1 and st["sampled"] == a["nreads"]), (name, r["bases"], st)
        assert sum(1 for per_rank in r["batches"] if per_rank) >= min(world, 2), name      # (more than one rank sampled something)


def test_damaged_gz_goes_back_to_the_one_sampler(tmp_path, monkeypatch):
    """A .gz file the chain of slices cannot finish - the CRC in its trailer is wrong, the file is cut short in the middle of its deflate
    data, bytes in its middle are damaged - must not hang a rank or give reads the reference would not: every rank agrees (status 1) and
    the caller reads the file with the ONE sampler on rank 0, which reports what gzip.open would (tests/test_reader.py).  A damaged
    member BEHIND the head-take is never looked at, as in the reference: status 0 and the reads of the sequential sampler."""
    import gzip
    import zlib
    from microbecensus_amd import _native
    inp = os.path.join(GOLD, "inputs")
    ex = gzip.open(os.path.join(inp, "example.fq.gz"), "rb").read()
    good = gzip.compress(ex, 1)
    bad_crc = bytearray(good); bad_crc[-8] ^= 0x55
    cut = good[: len(good) * 2 // 3]
    hurt = bytearray(good)
    for k in range(len(good) // 2, len(good) // 2 + 64):
        hurt[k] ^= 0xA5
    files = {"crc": bytes(bad_crc), "cut": cut, "hurt": bytes(hurt)}
    base = {"min_quality": -5, "mean_quality": -5, "max_unknown": 100, "filter_dups": False, "read_length": 100, "file_type": "fastq", "quality_offset": 32}
    cases = {}
    for name, blob in files.items():
        path = str(tmp_path / (name + ".fq.gz"))
        open(path, "wb").write(blob)
        cases[name] = dict(base, seqfiles=[path], nreads=10**9)
    cases["crc_behind_the_take"] = dict(base, seqfiles=[cases["crc"]["seqfiles"][0]], nreads=3000)
    cj = tmp_path / "cases.json"
    cj.write_text(json.dumps(cases))
    worker = tmp_path / "shard.py"
    worker.write_text(SHARD_WORKER.replace("assert D.sharded_sampling_usable(a)", "pass"))
    out = tmp_path / "shard.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MC_DIST_SLICE="150000", MC_DIST_GZ_CHUNK="65536")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
                           "--master-port", "29529", str(worker), REPO, str(cj), str(out)], env=env, timeout=600)
    res = json.load(open(out))
    for name in files:
        assert res[name]["status"] == 1, (name, res[name]["status"], res[name]["stats"])
    a = cases["crc_behind_the_take"]
    monkeypatch.setenv("MC_READER_REGION_BYTES", "200000")       # (the one sampler reads ahead by regions: small ones, so that it stops in front of the trailer too)
    monkeypatch.setenv("MC_READER_GZ_CHUNK", "65536")
    want, st = _native.sample_reads(a["seqfiles"], 100, 3000, True, 32, -5, -5, 100, False)
    r = res["crc_behind_the_take"]
    assert r["status"] == 0 and r["n_total"] == 3000 == len(want)
    at = 0
    for first, n, crc in sorted(tuple(b) for per_rank in r["batches"] for b in per_rank):
        assert first == at and crc == zlib.crc32(want[first:first + n].tobytes())
        at += n
    assert at == 3000 and all(r["stats"][k] == st[k] for k in ("too_short", "low_qual", "records"))


FAIL_WORKER = r'''
import json, os, sys
import numpy as np
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from microbecensus_amd import distributed as D, _native
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
a = json.load(open(sys.argv[2]))
out = {}
calls = [0]
def on_batch(block, first):
    calls[0] += 1
    if rank == 1 and calls[0] == 2:
        raise ValueError("search blew up on rank 1")
# (1) a sampler on every rank, the search of one rank fails: status 2 and ONE message on every rank - no silent second search
n_total, stats, bases, status = D.stream_batches_sharded(a, on_batch)
out["sharded"] = {"status": status, "error": stats.get("error")}
# (2) batches dealt by rank 0, the search of rank 1 fails at its second batch: every rank is handed the error, nobody waits for ever
calls[0] = 0
rd = None
if rank == 0:
    rd = _native.Reader(a["seqfiles"], a["read_length"], a["nreads"], False, 0, -5, -5, 100, False)
n_total, trace, err = D.stream_batches(rd, a["read_length"], on_batch)
if rd is not None:
    rd.close()
out["dealt"] = {"err": None if err is None else str(err)}
allo = [None] * world
dist.all_gather_object(allo, out)
if rank == 0:
    json.dump(allo, open(sys.argv[3], "w"))
dist.barrier()
dist.destroy_process_group()
'''


def test_a_failing_rank_is_agreed_on_by_all(tmp_path):
    """(a) stream_batches_sharded tells a failed rank (status 2 + the first failing rank's message, the same on every
    rank) from a ragged window (status 1: the only case run_pipeline_distributed falls back for); (b) stream_batches with a rank
    whose search raises ends on every rank with that error - the receiver's sentinel always reaches its searching thread."""
    import gzip
    plain = str(tmp_path / "metagenome.fa")
    open(plain, "wb").write(gzip.open(os.path.join(GOLD, "inputs", "metagenome.fa.gz"), "rb").read())
    a = {"min_quality": -5, "mean_quality": -5, "max_unknown": 100, "filter_dups": False, "seqfiles": [plain], "read_length": 100,
         "nreads": 10**9, "file_type": "fasta", "quality_offset": None}
    cj = tmp_path / "case.json"
    cj.write_text(json.dumps(a))
    worker = tmp_path / "fail.py"
    worker.write_text(FAIL_WORKER)
    out = tmp_path / "fail.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MC_DIST_SLICE="150000", MC_DIST_BATCH="2000")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                           "--master-port", "29531", str(worker), REPO, str(cj), str(out)], env=env, timeout=600)
    res = json.load(open(out))
    assert len(res) == 2
    for r in res:
        assert r["sharded"]["status"] == 2 and "rank 1" in r["sharded"]["error"] and "search blew up" in r["sharded"]["error"], r
        assert r["dealt"]["err"] and "search blew up" in r["dealt"]["err"] or "failed while searching" in (r["dealt"]["err"] or ""), r


DUPS_WORKER = SHARD_WORKER.replace("assert D.sharded_sampling_usable(a)", "assert D.sharded_dups_usable(a)").replace("D.stream_batches_sharded(a, on_batch)", "D.stream_batches_sharded_dups(a, on_batch)")


@pytest.mark.parametrize("world", [2, 3])
def test_every_rank_samples_its_own_slices_with_the_duplicate_filter(tmp_path, world):
    """stream_batches_sharded_dups (gloo, no GPU; VERDICT r05 "missing" #4 for -d): -d with a sampler on every rank - the ranks parse, filter and
    hash their own slices (100 KB here), exchange 32-byte descriptors and every rank walks the round's descriptors through its own copy of
    the set.  Kept reads in the order of their global indices, counters (duplicates included) and count_bases are those of the ONE
    sequential sampler (process_seqfile :328-367, :345, :354): the 300 bp library with -q 20 -d (the reference-made golden's input) in
    full and with the take ending inside a slice; two files whose second repeats reads of the first (exact and reverse complement, some
    of them after a first occurrence that failed the quality filter); a file with a record the reference raises at, behind the take (no
    error) and in front of it (status 3, the exception's name on every rank)."""
    import gzip
    import random
    import zlib
    from microbecensus_amd import _native
    inp = os.path.join(GOLD, "inputs")
    c5 = str(tmp_path / "c5_300bp.fq")
    open(c5, "wb").write(gzip.open(os.path.join(inp, "c5_300bp.fq.gz"), "rb").read())
    rng = random.Random(5 + world)
    comp = str.maketrans("ACGTN", "TGCAN")
    recs = []
    for i in range(3000):
        s = "".join(rng.choice("ACGT") for _ in range(rng.choice([90, 100, 100, 130])))
        lo = rng.choice([2, 25, 30])                              # a third of the records fail -q 20
        q = "".join(chr(33 + rng.randrange(lo, 41)) for _ in s)
        recs.append((s, q))
    second = []
    for i in range(3000):
        s, q = rng.choice(recs)
        if rng.random() < 0.5:
            s = s[::-1].translate(comp)
        if rng.random() < 0.3:
            s, q = "".join(rng.choice("ACGT") for _ in range(100)), "I" * 100
        second.append((s, "".join(chr(33 + rng.randrange(25, 41)) for _ in s)))
    fa, fb, fbad = str(tmp_path / "a.fq"), str(tmp_path / "b.fq"), str(tmp_path / "bad.fq")
    for path, rr in ((fa, recs), (fb, second)):
        with open(path, "w") as f:
            f.write("".join("@r%d\n%s\n+\n%s\n" % (i, s, q) for i, (s, q) in enumerate(rr)))
    bad = list(recs)
    bad[2000] = ("ACGTX" * 20, "I" * 100)
    with open(fbad, "w") as f:
        f.write("".join("@r%d\n%s\n+\n%s\n" % (i, s, q) for i, (s, q) in enumerate(bad)))
    base = {"min_quality": -5, "mean_quality": -5, "max_unknown": 100, "filter_dups": True, "file_type": "fastq", "quality_offset": 33}
    cases = {
        "c5": dict(base, seqfiles=[c5], read_length=300, nreads=10**9, min_quality=20, quality_offset=32 + 1),
        "c5_take": dict(base, seqfiles=[c5], read_length=300, nreads=2500, min_quality=20),
        "two_files": dict(base, seqfiles=[fa, fb], read_length=90, nreads=10**9, min_quality=20, mean_quality=25),
        "two_files_take": dict(base, seqfiles=[fa, fb], read_length=90, nreads=2600, min_quality=20, mean_quality=25),
        "bad_behind_the_take": dict(base, seqfiles=[fbad], read_length=90, nreads=700, min_quality=20),
        "bad_reached": dict(base, seqfiles=[fbad], read_length=90, nreads=10**9, min_quality=20),
    }
    cj = tmp_path / "cases.json"
    cj.write_text(json.dumps(cases))
    worker = tmp_path / "dups.py"
    worker.write_text(DUPS_WORKER)
    out = tmp_path / "dups.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MC_DIST_SLICE="100000")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                           "--master-port", str(29561 + world), str(worker), REPO, str(cj), str(out)], env=env, timeout=900)
    res = json.load(open(out))
    for name, a in cases.items():
        r = res[name]
        if name == "bad_reached":
            assert r["status"] == 3 and "KeyError" in r["stats"]["error"], r
            with pytest.raises(_native.ReferenceError_):
                _native.sample_reads(a["seqfiles"], a["read_length"], a["nreads"], True, a["quality_offset"], a["min_quality"], a["mean_quality"], a["max_unknown"], True)
            continue
        want, st = _native.sample_reads(a["seqfiles"], a["read_length"], a["nreads"], True, a["quality_offset"], a["min_quality"], a["mean_quality"], a["max_unknown"], True)
        assert r["status"] == 0 and r["n_total"] == st["sampled"] == len(want), (name, r["n_total"], st)
        at = 0
        for first, n, crc in sorted(tuple(b) for per_rank in r["batches"] for b in per_rank):
            assert first == at and crc == zlib.crc32(want[first:first + n].tobytes()), (name, first, at)
            at += n
        assert at == len(want), name
        for k in ("too_short", "low_qual", "dups", "records"):
            assert r["stats"][k] == st[k], (name, k, r["stats"], st)
        assert r["bases"] == (st["bases"] if st["exhausted"] else -1) or (r["bases"] == -1 and st["sampled"] == a["nreads"]), (name, r["bases"], st)
        assert sum(1 for per_rank in r["batches"] if per_rank) >= 2, name
    assert res["two_files"]["stats"]["dups"] > 500 and res["c5"]["stats"]["dups"] > 50
