"""BASELINE configs[1], [3] and [4] on the GPU: small versions against goldens produced by RUNNING THE REFERENCE here
(tests/golden/make_golden.py: 100 bp FASTA; a paired 150 bp library given as two files; 300 bp FASTQ with -q 20 -d), and the
full-size shape of configs[1] (2.3 M reads of 100 bp: crosses the 2,097,151-read batch limit) through a size-independent
property.  All calls go through the C ABI."""
import gzip
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from microbecensus_amd import microbe_census as mc

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
GOLD = os.path.join(HERE, "golden")
INPUTS = os.path.join(GOLD, "inputs")

CASES = {
    "c2_100bp": {"seqfiles": ["c2_100bp.fa.gz"], "threads": 8},
    "c4_paired": {"seqfiles": ["c4_pair_1.fq.gz", "c4_pair_2.fq.gz"], "threads": 8, "nreads": 20000},
    "c5_300bp_q20_dups": {"seqfiles": ["c5_300bp.fq.gz"], "threads": 8, "min_quality": 20, "filter_dups": True},
    # a phred+64 FASTQ through the quality filter: the offset is DETECTED here (mc_quality_offset; the first record decides nothing),
    # as the reference detected it for the golden (tests/golden/make_phred64_golden.py; microbe_census.py:175-187, :265-279)
    "c1_phred64_q_m": {"seqfiles": ["phred64_110bp.fq.gz"], "threads": 8, "read_length": 100, "min_quality": 10, "mean_quality": 25},
}


def _args(case):
    a = dict(CASES[case])
    a["seqfiles"] = [os.path.join(INPUTS, f) for f in a["seqfiles"]]
    return a


# (the RCCL test first: a multi-GPU box that runs this file reaches it whatever happens later)
def _two_ranks(tmp_path, backend, port, world=2, batch="3000", gz_sharded=False):
    worker = tmp_path / "w.py"
    worker.write_text(r'''
import json, os, sys
sys.path.insert(0, sys.argv[1])
import torch
import torch.distributed as dist
from microbecensus_amd import distributed as D
backend = sys.argv[3]
local = int(os.environ["LOCAL_RANK"])
if backend == "nccl":
    torch.cuda.set_device(local)
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
else:
    dist.init_process_group(backend="gloo")
inp = os.path.join(sys.argv[1], "tests", "golden", "inputs")
est, args = D.run_pipeline_distributed({"seqfiles": [os.path.join(inp, "c4_pair_1.fq.gz"), os.path.join(inp, "c4_pair_2.fq.gz")], "nreads": 20000},
                                       device=local if backend == "nccl" else 0)
got = [None] * dist.get_world_size()
dist.all_gather_object(got, D.run_pipeline_distributed.last_batches)
if dist.get_rank() == 0:
    tr = D.run_pipeline_distributed.last_trace
    json.dump({"est": est, "sampled": args["sampled_reads"], "L": args["read_length"], "world": dist.get_world_size(), "backend": dist.get_backend(),
               "deals": -1 if tr is None else sum(1 for t in tr if t[0] == "deal"), "batches_per_rank": got}, open(sys.argv[2], "w"))
dist.barrier()
dist.destroy_process_group()
''')
    out = tmp_path / "o.json"
    # MC_DIST_GZ=0: rank 0 samples the .gz files and DEALS batches (20,000 reads in 7 batches of 3,000); gz_sharded: every rank decodes its own chunks
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", MC_DIST_BATCH=batch, MC_DIST_GZ="1" if gz_sharded else "0")
    if gz_sharded:
        env.update(MC_DIST_GZ_CHUNK="65536", MC_DIST_SLICE="150000")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                           "--master-port", str(port), str(worker), REPO, str(out), backend], env=env, timeout=900)
    return json.load(open(out))


def test_config4_shape_two_ranks_rccl(tmp_path):
    """The same over RCCL (backend "nccl") with one GPU per rank: batches dealt GPU to GPU (isend / recv), all_reduce of the
    per-family sums.  Needs two GPUs."""
    from microbecensus_amd import _native
    if _native.load_library().mc_device_count() < 2:              # (asked of the HIP library this process already uses, not of torch)
        pytest.skip("needs two GPUs (the driver's multi-GPU node)")
    g = json.load(open(os.path.join(GOLD, "c4_paired.json")))
    res = _two_ranks(tmp_path, "nccl", 29543)
    assert res["backend"] == "nccl" and res["world"] == 2 and res["sampled"] == g["sampled_reads"]
    assert abs(res["est"] - g["est_ags"]) <= 1e-9 * g["est_ags"]


def test_rccl_first_contact_on_one_gpu(tmp_path):
    """What a one-GPU box can show of RCCL before the driver's node runs the two-rank test above: the library of this image loads in a
    rank process (HSA_ENABLE_IPC_MODE_LEGACY=0), a communicator of world size 1 forms on the GPU, the all_reduce the product issues -
    int64 SUM over the per-family accumulator vector, in HBM - returns the vector, and run_pipeline_distributed under backend "nccl"
    (device tensors, eng.attach path switched on) gives the reference's AGS for the paired library."""
    worker = tmp_path / "w1.py"
    worker.write_text(r'''
import json, os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
import torch
import torch.distributed as dist
from microbecensus_amd import distributed as D, _native
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
nf = len(_native.load_model()["families"])
rng = np.random.default_rng(5)
acc = (rng.integers(0, 1 << 40, nf), rng.integers(0, 1 << 50, nf), rng.integers(0, 1 << 40, (nf, D.MAX_TARGET_LEN)))
flat = np.concatenate([a.ravel() for a in acc]).astype(np.int64)
t = torch.from_numpy(flat).to("cuda:0")
dist.all_reduce(t, op=dist.ReduceOp.SUM)
same = bool((t.cpu().numpy() == flat).all())
inp = os.path.join(sys.argv[1], "tests", "golden", "inputs")
est, args = D.run_pipeline_distributed({"seqfiles": [os.path.join(inp, "c4_pair_1.fq.gz"), os.path.join(inp, "c4_pair_2.fq.gz")], "nreads": 20000}, device=0)
json.dump({"same": same, "words": int(flat.size), "est": est, "sampled": args["sampled_reads"], "backend": dist.get_backend(),
           "version": ".".join(str(v) for v in torch.cuda.nccl.version())}, open(sys.argv[2], "w"))
dist.barrier()
dist.destroy_process_group()
''')
    out = tmp_path / "o1.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29561", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    subprocess.check_call([sys.executable, str(worker), REPO, str(out)], env=env, timeout=900)
    res = json.load(open(out))
    g = json.load(open(os.path.join(GOLD, "c4_paired.json")))
    assert res["backend"] == "nccl" and res["same"] and res["words"] > 1000, res
    assert res["sampled"] == g["sampled_reads"] and abs(res["est"] - g["est_ags"]) <= 1e-9 * g["est_ags"]
    print("RCCL", res["version"])


def test_config4_shape_two_ranks_gloo(tmp_path):
    """The paired library given as `a,b` through run_pipeline_distributed with two ranks (both on this box's GPU, gloo): rank 0
    samples and deals batches to the two ranks while it samples, the reduced per-family sums give the reference's AGS for the same pair."""
    g = json.load(open(os.path.join(GOLD, "c4_paired.json")))
    res = _two_ranks(tmp_path, "gloo", 29541)
    assert res["world"] == 2 and res["sampled"] == g["sampled_reads"] and res["L"] == 150
    assert res["deals"] == 7                                      # streamed: 3,000-read batches dealt round robin while the sampler runs
    assert abs(res["est"] - g["est_ags"]) <= 1e-9 * g["est_ags"]


def test_config4_shape_two_ranks_gz_decoded_on_both_gloo(tmp_path):
    """The same paired .gz library with a sampler on EVERY rank (round 6): the ranks decode their own chunks of the gzip members (64 KB chunks,
    two per rank and round here), hand the 32 KB windows along, sample the records that start in their text - nothing is dealt by rank 0 -
    and the reduced sums give the reference's AGS."""
    g = json.load(open(os.path.join(GOLD, "c4_paired.json")))
    res = _two_ranks(tmp_path, "gloo", 29555, gz_sharded=True)
    assert res["world"] == 2 and res["sampled"] == g["sampled_reads"] and res["L"] == 150 and res["deals"] == -1
    assert min(res["batches_per_rank"]) >= 1
    assert abs(res["est"] - g["est_ags"]) <= 1e-9 * g["est_ags"]


def test_config4_shape_eight_ranks_gloo(tmp_path):
    """World size 8 - what the driver's multi-GPU node runs - before any hardware has it: eight ranks on this box's
    one GPU (gloo), the reference's paired golden dealt in batches of 1,500 reads: the reference's AGS (the 'cov' sums are finished from
    exact integer sums in another order than the reference's running sum: 1e-9 relative, as the two-rank test), 14 deals, and every rank
    searched at least one batch (credits: two per rank, the dealer goes round the ranks)."""
    g = json.load(open(os.path.join(GOLD, "c4_paired.json")))
    res = _two_ranks(tmp_path, "gloo", 29551, world=8, batch="1500")
    assert res["world"] == 8 and res["sampled"] == g["sampled_reads"] and res["L"] == 150
    assert res["deals"] == 14 and sum(res["batches_per_rank"]) == 14 and min(res["batches_per_rank"]) >= 1, res
    assert abs(res["est"] - g["est_ags"]) <= 1e-9 * g["est_ags"]


def test_bench_eight_ranks_on_one_gpu_equals_one_engine(tmp_path):
    """bench.py --gpus 8 as the driver will launch it, with gloo and all ranks on this box's GPU: the line's `rccl`
    object must show eight ranks (rank_sum 36), and the reduced per-family vector - hits, alignment sums, alignment sums per target
    length, summed over the ranks by the collective - must be the vector ONE engine computes for the same 4 M reads of the paired
    library (crc32 of the int64 arrays), i.e. no read searched twice or dropped by the sharding, no sum lost in the reduce."""
    import zlib
    import numpy as np
    from microbecensus_amd import _native, distributed as mcd, synth
    B, K, W, L = 250_000, 2, 8, 150
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(W), "--backend", "gloo", "--batch", str(B), "--steps", str(K), "--warmup", "1",
                          "--no-cpu-baseline", "--e2e-reads", "2000000", "--c5-reads", "0"], env=env, stdout=subprocess.PIPE, timeout=1500, check=True).stdout.decode()
    line = json.loads([l for l in out.splitlines() if l.startswith('{"metric"')][-1])
    r = line["rccl"]
    assert line["n_gpus"] == W and r["world_size"] == W and r["rank_sum"] == 36 == r["rank_sum_expected"] and r["backend"] == "gloo"
    assert line["e2e"]["sampled_reads"] == 2_000_000 and line["e2e"]["est_ags"] > 1e6
    gz = line["e2e"]["gz"]                                        # the FASTQ.gz of eight members, inflated by all ranks: every read sampled, nothing dealt
    assert gz["sampled_reads"] == gz["reads"] == 2_000_000 and "nothing dealt by rank 0: True" in gz["what"] and abs(gz["est_ags"] / line["e2e"]["est_ags"] - 1) < 0.02
    # the same reads through one engine of this process: file 1 (mate 1 of every fragment), then file 2
    model = _native.load_model()
    fams = model["families"]
    nf = len(fams)
    gen = synth.GenomeReads(device="cuda:0", seed=20261001)
    F = B * K * W // 2
    eng = _native.Engine(device=0)
    try:
        eng.set_run(L, model["pars"][str(L)], fams)
        hits, aln, bylen = np.zeros(nf, np.int64), np.zeros(nf, np.int64), np.zeros((nf, mcd.MAX_TARGET_LEN), np.int64)
        for mate in (0, 1):
            for lo in range(0, F, 1_000_000):
                n = min(1_000_000, F - lo)
                reads = gen.paired(n, L, frag=300, first=lo)[mate]
                eng.attach(reads.data_ptr(), n)
                eng.run_range(0, n, first_read_id=0)
                h, a, b = mcd.family_accumulators(eng.best_hits(), nf)
                hits += h; aln += a; bylen += b
                eng.attach(0, 0)
    finally:
        eng.close()
    assert line["config"]["classified_reads"] == int(hits.sum()) > 1000
    assert r["reduced_vector_crc32"] == zlib.crc32(np.concatenate([hits, aln, bylen.ravel()]).tobytes())


def test_config5_shape_two_ranks_sharded_duplicate_filter_gloo(tmp_path):
    """BASELINE configs[4]'s shape (300 bp FASTQ, -q 20 -d) through run_pipeline_distributed on a PLAIN file with two ranks (both on this
    box's GPU, gloo): with -d, too, every rank samples its own slices - descriptors exchanged, the duplicate verdicts walked on every rank
    (stream_batches_sharded_dups; slices of 400 KB here) - and nothing is dealt by rank 0: the reference's sample size and AGS for the
    golden made by the reference itself (tests/golden/c5_300bp_q20_dups.json)."""
    g = json.load(open(os.path.join(GOLD, "c5_300bp_q20_dups.json")))
    fq = tmp_path / "c5_300bp.fq"
    fq.write_bytes(gzip.open(os.path.join(INPUTS, "c5_300bp.fq.gz"), "rb").read())
    worker = tmp_path / "w.py"
    worker.write_text(r'''
import json, os, sys
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from microbecensus_amd import distributed as D
dist.init_process_group(backend="gloo")
est, args = D.run_pipeline_distributed({"seqfiles": [sys.argv[3]], "min_quality": 20, "filter_dups": True}, device=0)
if dist.get_rank() == 0:
    json.dump({"est": est, "sampled": args["sampled_reads"], "L": args["read_length"], "dealt": D.run_pipeline_distributed.last_trace is not None,
               "stats": D.run_pipeline_distributed.last_stats}, open(sys.argv[2], "w"))
dist.barrier()
dist.destroy_process_group()
''')
    out = tmp_path / "o.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", MC_DIST_SLICE="400000")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                           "--master-port", "29553", str(worker), REPO, str(out), str(fq)], env=env, timeout=900)
    res = json.load(open(out))
    assert res["sampled"] == g["sampled_reads"] and res["L"] == 300 and res["dealt"] is False and res["stats"]["dups"] > 0
    assert abs(res["est"] - g["est_ags"]) <= 1e-9 * g["est_ags"]


def test_sharded_sampling_two_ranks_gloo(tmp_path):
    """run_pipeline_distributed on a PLAIN file with two ranks (both on this box's GPU, gloo): every rank samples its own slices
    (mc_reader_open_range; slices of 1 MB here), the head-take and the read indices come from the exchanged counts, one all_reduce of
    the per-family sums - the reference's AGS for the unit-test metagenome, and nothing dealt by rank 0."""
    import gzip
    import subprocess
    import sys
    g = json.load(open(os.path.join(GOLD, "unittest_metagenome.json")))
    fa = tmp_path / "metagenome.fa"
    fa.write_bytes(gzip.open(os.path.join(INPUTS, "metagenome.fa.gz"), "rb").read())
    worker = tmp_path / "w.py"
    worker.write_text(r'''
import json, os, sys
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from microbecensus_amd import distributed as D
dist.init_process_group(backend="gloo")
est, args = D.run_pipeline_distributed({"seqfiles": [sys.argv[3]]}, device=0)
if dist.get_rank() == 0:
    json.dump({"est": est, "sampled": args["sampled_reads"], "L": args["read_length"], "dealt": D.run_pipeline_distributed.last_trace is not None,
               "stats": D.run_pipeline_distributed.last_stats}, open(sys.argv[2], "w"))
dist.barrier()
dist.destroy_process_group()
''')
    out = tmp_path / "o.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", MC_DIST_SLICE="1000000")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                           "--master-port", "29547", str(worker), REPO, str(out), str(fa)], env=env, timeout=900)
    res = json.load(open(out))
    assert res["sampled"] == g["sampled_reads"] and res["L"] == 100 and res["dealt"] is False and res["stats"]["records"] == g["sampled_reads"]
    assert abs(res["est"] - g["est_ags"]) <= 1e-9 * g["est_ags"]


@pytest.mark.parametrize("case", sorted(CASES))
def test_stage_by_stage_against_the_reference(case):
    """sampler -> search -> classification -> aggregation -> estimate: the temp FASTA, the m8 file, the best hits, the
    per-family sums and the AGS the reference produced for the same input."""
    g = json.load(open(os.path.join(GOLD, case + ".json")))
    args = _args(case)
    paths = mc.get_relative_paths(args)
    mc.check_paths(paths); mc.check_input(args); mc.impute_missing_args(args); mc.check_arguments(args)
    assert args["read_length"] == g["args"]["read_length"]
    mc.process_seqfile(args, paths)
    assert args["sampled_reads"] == g["sampled_reads"]
    assert hashlib.md5(open(paths["tempfile"], "rb").read()).hexdigest() == g["reads_md5"]
    mc.search_seqs(args, paths)
    m8 = b"".join(l for l in open(paths["tempfile"] + ".m8", "rb") if not l.startswith(b"#"))
    assert m8.count(b"\n") == g["m8_rows"]
    assert hashlib.md5(m8).hexdigest() == g["m8_md5"]
    assert m8 == gzip.open(os.path.join(GOLD, case + ".m8.gz"), "rb").read()
    best = mc.classify_reads(args, paths)
    assert best == g["best_hits"]
    agg = mc.aggregate_hits(args, paths, best)
    assert agg == g["agg_hits"]
    mc.clean_up(paths)
    assert mc.estimate_average_genome_size(args, paths, agg) == g["est_ags"]


@pytest.mark.parametrize("case", sorted(CASES))
def test_run_pipeline_fused_path(case):
    """run_pipeline (sampler beside the search, mc_search_files): the reference's AGS, bit for bit."""
    g = json.load(open(os.path.join(GOLD, case + ".json")))
    est, args = mc.run_pipeline(_args(case))
    assert args["sampled_reads"] == g["sampled_reads"] and est == g["est_ags"]


def test_run_pipeline_over_several_handles_in_one_process(monkeypatch):
    """args['devices'] (the optional GPU-count key): the sampler's batches dealt to several engines of this process
    (mc_search_files_multi) - here three handles on this box's one GPU and batches of 4,000 reads - must give the reference's AGS,
    bit for bit, like one engine does."""
    g = json.load(open(os.path.join(GOLD, "c2_100bp.json")))
    monkeypatch.setenv("MC_STREAM_BATCH", "4000")
    args = _args("c2_100bp")
    args.pop("device", None)
    args["devices"] = [0, 0, 0]
    args["nreads"] = 10_000_000                                  # (else a run of this size would be given one device)
    est, out = mc.run_pipeline(args)
    assert out["sampled_reads"] == g["sampled_reads"] and est == g["est_ags"]
    assert sorted(k for k in mc._engines if isinstance(k, tuple)) == [(0, 1), (0, 2)]     # (the extra handles are opened side by side: any order)


def test_config2_full_size_batch_split_invariance():
    """2.3 M synthetic 100 bp reads (more than one 2,097,151-read batch): what mc_search returns for the whole set equals what it
    returns for two halves searched separately with offset read ids - rows, best hits, and a checksum over every field."""
    from microbecensus_amd import _native, synth
    gen = synth.GenomeReads(device="cpu", seed=7)               # (torch stays off the GPU in this process: the HIP library owns it)
    n, L = 2_300_000, 100
    reads = gen.single(n, L).numpy()
    eng = _native.Engine(device=0)
    try:
        model = _native.load_model()
        eng.set_run(L, model["pars"][str(L)], model["families"])
        rows, best = eng.search(reads)
        cut = 1_234_567
        r1, b1 = eng.search(reads[:cut])
        r2, b2 = eng.search(reads[cut:], first_read_id=cut)
    finally:
        eng.close()
    assert len(rows) == len(r1) + len(r2) and len(rows) > 1_000_000
    both = np.concatenate([r1, r2])
    for f in rows.dtype.names:
        assert (rows[f] == both[f]).all(), f
    bb = np.concatenate([b1, b2])
    for f in best.dtype.names:
        assert (best[f] == bb[f]).all(), f
    assert (np.diff(rows["query"]) >= 0).all()                    # m8 order: ascending read id
    assert rows["query"].max() < n and best["read"].max() < n and len(best) > 5000


def test_multi_device_entry_deals_batches(monkeypatch):
    """mc_search_files_multi with two handles (both on this box's GPU), batches of 9,000 reads dealt between them: the union of
    their best hits is what one handle finds."""
    from microbecensus_amd import _native
    monkeypatch.setenv("MC_STREAM_BATCH", "9000")
    path = os.path.join(INPUTS, "metagenome.fa.gz")
    model = _native.load_model()
    engs = [_native.Engine(device=0) for _ in range(2)]
    try:
        for e in engs:
            e.set_run(100, model["pars"]["100"], model["families"])
        rd = _native.Reader([path], 100, 1000000, False, 0, -5, -5, 100, False)
        both = _native.search_files_multi(engs, rd)
        assert min(e.stats()["reads"] for e in engs) > 9000          # both handles really worked
        rd.close()
        rd = _native.Reader([path], 100, 1000000, False, 0, -5, -5, 100, False)
        rows, one = engs[0].search_files(rd, keep_rows=False)
        rd.close()
    finally:
        for e in engs:
            e.close()
    g = json.load(open(os.path.join(GOLD, "unittest_metagenome.json")))
    assert len(both) == len(one) == len(g["best_hits"])
    for f in one.dtype.names:
        assert (both[f] == one[f]).all(), f


def test_rapsearch_compatible_executable(tmp_path):
    """scripts/rapsearch_mi355x behind RAPsearch2's command line (the reference's -r hook and training/search_reads.py:57 call it
    like this): -h passes check_rapsearch, the m8 has RAPsearch2's five header lines and the reference's rows, the .aln exists
    (empty for -b 0); run_pipeline with args['rapsearch'] = the executable gives the reference's AGS."""
    import contextlib
    import io
    exe = os.path.join(REPO, "scripts", "rapsearch_mi355x")
    mc.check_rapsearch(exe)
    g = json.load(open(os.path.join(GOLD, "config1_example_fq.json")))
    fa = tmp_path / "reads.fa"
    fa.write_bytes(gzip.open(os.path.join(GOLD, "config1_example_fq.reads.fa.gz"), "rb").read())
    db = mc._rapdb_for_external_search()
    out = str(tmp_path / "out")
    subprocess.check_call([exe, "-q", str(fa), "-d", db, "-o", out, "-z", "1", "-e", "1", "-t", "n", "-p", "f", "-b", "0"], stdout=subprocess.DEVNULL)
    lines = open(out + ".m8", "rb").readlines()
    assert [l[:1] for l in lines[:6]] == [b"#"] * 5 + [b"1"] and lines[0] == b"# RAPSearch\n" and lines[4].startswith(b"# Fields: Query\tSubject\tidentity")
    assert hashlib.md5(b"".join(lines[5:])).hexdigest() == g["m8_md5"]
    assert os.path.getsize(out + ".aln") == 0
    args = {"seqfiles": [os.path.join(INPUTS, "example.fq.gz")], "nreads": 10000, "read_length": 100, "threads": 1, "rapsearch": exe}
    with contextlib.redirect_stdout(io.StringIO()):
        est, args = mc.run_pipeline(args)
    assert est == g["est_ags"]


def test_training_grid_on_device_rows():
    """The training workflow's grid search (training/training.py:311-334; 4 aln_covs x 6 max_pids x 27 min_scores as
    training/class_reads.py:51-53 sets them) on the device rows of the unit-test metagenome against the .hits table the
    REFERENCE's own functions produced from its own m8 (tests/golden/training_grid_unittest.json.gz, written by
    tests/golden/make_training_golden.py, which executes training.py:229-334 unchanged): hits and aligned residues identical,
    coverage sums to 1e-12."""
    from microbecensus_amd import _native
    g = json.load(open(os.path.join(GOLD, "unittest_metagenome.json")))
    gold = json.load(gzip.open(os.path.join(GOLD, "training_grid_unittest.json.gz"), "rt"))
    L = 100
    model = _native.load_model()
    fams = model["families"]
    aln_covs, max_pids, min_scores = gold["aln_covs"], gold["max_pids"], gold["min_scores"]
    assert (len(aln_covs), len(max_pids), len(min_scores)) == (4, 6, 27)
    want = {}
    for fam, aln_cov, max_pid, min_score, hits, aln, cov in gold["rows"]:
        want[(aln_covs.index(aln_cov), max_pids.index(max_pid), min_scores.index(min_score), fams.index(fam))] = (hits, aln, cov)
    assert len(want) == gold["n_rows_with_hits"]
    # --- the device grid on the device rows
    reads, st = _native.sample_reads([os.path.join(INPUTS, "metagenome.fa.gz")], L, 1000000, False, 0, -5, -5, 100, False)
    eng = _native.Engine(device=0)
    try:
        eng.set_run(L, model["pars"][str(L)], fams)
        rows, _ = eng.search(reads)
        assert len(rows) == g["m8_rows"]
        gh, ga, gc = eng.grid_classify(aln_covs, max_pids, min_scores)
    finally:
        eng.close()
    assert gh.sum() == sum(w[0] for w in want.values()) and gh.sum() > 5000
    for k, w in want.items():
        assert gh[k] == w[0] and ga[k] == w[1], k
        assert abs(gc[k] - w[2]) <= 1e-12 * w[2], k
    assert int((gh > 0).sum()) == len(want)


def test_config5_shape_at_size_native_sampler_equals_the_reference(tmp_path):
    """BASELINE configs[4] shape at a size where the parallel reader cuts many regions and pieces: 150,000 FASTQ records of 300 bp
    (tests/golden/c5_at_size.py) with -q 20 -d.  The expected sample comes from the REFERENCE's own process_seqfile run on the same
    file (tests/golden/c5_at_size.json, make_sampler_at_size_golden.py): counters, sample size and the md5 of the temp FASTA; the
    fused path (native sampler beside the search) and the stage-by-stage path then give the same AGS."""
    sys.path.insert(0, GOLD)
    import c5_at_size
    g = json.load(open(os.path.join(GOLD, "c5_at_size.json")))
    fq = tmp_path / "c5.fq"
    c5_at_size.write_fastq(str(fq))
    assert hashlib.md5(fq.read_bytes()).hexdigest() == g["file_md5"]
    base = {"seqfiles": [str(fq)], "min_quality": 20, "filter_dups": True, "nreads": 10_000_000}
    est, args = mc.run_pipeline(dict(base))
    assert args["read_length"] == g["read_length"] and args["quality_offset"] == g["quality_offset"] and args["sampled_reads"] == g["sampled_reads"]
    # stage by stage: the temp FASTA is the reference's, byte for byte
    args2 = dict(base)
    paths = mc.get_relative_paths(args2)
    mc.check_input(args2); mc.impute_missing_args(args2); mc.check_arguments(args2)
    mc.process_seqfile(args2, paths)
    assert hashlib.md5(open(paths["tempfile"], "rb").read()).hexdigest() == g["reads_md5"]
    mc.search_seqs(args2, paths)
    est2 = mc.estimate_average_genome_size(args2, paths, mc.aggregate_hits(args2, paths, mc.classify_reads(args2, paths)))
    mc.clean_up(paths)
    assert est == est2
    assert mc.count_bases(args2) == g["count_bases"]


@pytest.mark.gpu
def test_read_lengths_off_the_reference_grid_against_the_oracle(tmp_path):
    """The C ABI takes any read length from 18 to 510 bp (the reference's CLI only the 20 lengths of its parameter table):
    the shortest, some odd ones and the longest, genome reads, m8 md5 against the oracle's (every launch shape of the seed
    kernel and both staging forms of k_translate_seg are among them)."""
    from microbecensus_amd import _native, synth
    port, db = os.path.join(REPO, "oracle", "rs_port"), os.path.join(REPO, "oracle", "_ref", "rapdb_2.15")
    assert os.path.exists(port) and os.path.exists(db), "oracle not built (tests/conftest.py fails every GPU test without it)"
    gen = synth.GenomeReads(device="cpu", seed=77)
    eng = _native.Engine(device=0)
    try:
        for L, n in ((18, 4000), (33, 6000), (47, 6000), (64, 6000), (199, 5000), (255, 4000), (510, 2500)):
            reads = gen.single(n, L, first=L * 7919).numpy()
            fa = tmp_path / "r.fa"
            fa.write_text("".join(">%d\n%s\n" % (i, bytes(r).decode()) for i, r in enumerate(reads)))
            eng.set_run(L)
            eng.search(reads)
            eng.write_m8(str(tmp_path / "gpu.m8"))
            subprocess.check_call([port, db, str(fa), str(tmp_path / "cpu.m8")])
            assert hashlib.md5((tmp_path / "gpu.m8").read_bytes()).hexdigest() == hashlib.md5((tmp_path / "cpu.m8").read_bytes()).hexdigest(), "read length %d" % L
    finally:
        eng.close()
