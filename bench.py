#!/usr/bin/env python3
"""bench.py - reads/s of the MI355X translated-search hot path on synthetic shotgun reads.

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no torchrun environment, bench.py starts the N ranks itself (a child `python -m torch.distributed.run
--nnodes=1 --nproc-per-node N ... bench.py ...`, started before anything touches the GPU), relays rank 0's JSON line and
exits with the child's code.  Launched under torchrun (RANK / LOCAL_RANK / WORLD_SIZE set) it is one rank.

Workload: error-free reads sampled from the reference's 30 real genomes (tests/golden/genomes, 84.8 Mbp,
251 contigs), uniform start, strand by fair coin, resident in HBM before the timed region starts.
  N = 1: BASELINE configs[2] - K x batch distinct single reads of --read-len bp (defaults: 10 x 2 M = 20 M reads of 150 bp).
  N > 1: BASELINE configs[3] in weak-scaling form - a paired library (mate 2 = reverse complement of the fragment end, insert
         --insert), the stream "all of file 1, then all of file 2" cut into contiguous blocks, one per rank, K x batch reads each.
One "step" is one pass of the whole device pipeline (translate + SEG, seeds, extension, ranking, classification) over one batch;
the per-family accumulators of every step are summed over the ranks with an RCCL all_reduce.

Rank 0 prints ONE JSON line: metric / value (whole-job reads/s), `roofline` for the dominant kernel and, at N = 1, `cpu_baseline`
(the reference's own RAPsearch2 binary on this host's cores on bounded samples of the same reads, m8 compared by md5) and `e2e`
(file -> AGS through run_pipeline).
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

METRIC = "reads/sec searched vs marker DB + AGS abs-error, 150 bp @ 1/2/4/8 GPU"
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
def seed_positions(L):
    """seed positions of a read's six frames (a frame of n residues has n - 6)"""
    return 2 * sum(max((L - r) // 3 - 6, 0) for r in range(3))


SURVEY_A = {100: 127403, 150: 207923, 300: 452893}   # modelled algorithmic bytes per read of the whole path


def spawn_ranks(n, argv):
    """Start n ranks (one per GPU) as a child torchrun; nothing in this process has touched the GPU yet."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + argv
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in p.stdout:
        if ln.startswith('{"metric"'):
            line = ln
        else:
            sys.stderr.write(ln)
    rc = p.wait()
    if line:
        sys.stdout.write(line)
        sys.stdout.flush()
    return rc if rc else (0 if line else 1)


STAGE_KERNELS = {   # kernels whose HIP-event time a stage of mc_stats spans (prefixes of the names in the rocprofv3 summaries)
    "k_translate_seg": ["k_translate_seg"], "k_enumerate_t0": ["k_enumerate_t0<"], "k_enumerate": ["k_enumerate_t0<", "k_enumerate"], "k_eval_seeds": ["k_eval_seeds"],
    "k_gapped": ["k_gap_dedupe", "k_gapped_lds", "k_gapped", "k_gap_emit"], "k_finish": ["k_finish", "k_finish_heavy", "k_heavy_lists", "k_emit_rows"],
    "sort": ["k_make_keys", "k_gather", "k_heads"],
}


def profiled_traffic(stage, n_batch):
    """HBM-side bytes per launch of the kernels of `stage` from the newest committed rocprofv3 PMC summary
    (profiles/rNN_hbm_traffic.json: FETCH_SIZE and WRITE_SIZE collected in their own --pmc passes; read bytes = 2 x FETCH_SIZE on
    gfx950, see profiles/README.md), scaled from the profiled batch to this run's batch.  None if no profile is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_hbm_traffic.json")))
    if not files:
        return None
    try:
        tr = json.load(open(files[-1]))
        line = json.load(open(files[-1].replace("_hbm_traffic.json", "_bench_line.json")))
        per = float(line["config"]["batch"])
        total, found = 0.0, False
        for k, t in tr.items():
            if (k.startswith("k_enumerate") and k.rstrip().endswith("true>")) or t.get("write_kib_per_launch") is None:      # (the counting form of the seed kernel is not the timed one)
                continue
            if any(k == p or k.startswith(p + "<") or (p.endswith("<") and k.startswith(p)) for p in STAGE_KERNELS.get(stage, [stage])):
                total += (2.0 * t["fetch_kib_per_launch"] + t["write_kib_per_launch"]) * 1024.0
                found = True
        return total / per * n_batch if found else None
    except Exception:
        return None


def profiled_l2_hit_rate(stage):
    """TCC_HIT / (TCC_HIT + TCC_MISS) of the stage's kernels in the newest committed PMC summary (profiles/rNN_pmc_per_launch.csv)."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_per_launch.csv")))
    if not files:
        return None
    try:
        hit = miss = 0.0
        for r in csv.DictReader(open(files[-1])):
            k = r["Kernel"]
            if (k.startswith("k_enumerate") and k.rstrip().rstrip('"').endswith("true>")) or not r.get("TCC_HIT_sum"):
                continue
            if any(k == p or k.startswith(p + "<") or (p.endswith("<") and k.startswith(p)) for p in STAGE_KERNELS.get(stage, [stage])):
                hit += float(r["TCC_HIT_sum"]); miss += float(r["TCC_MISS_sum"])
        return round(hit / (hit + miss), 4) if hit + miss > 0 else None
    except Exception:
        return None


def ags_abs_error(device):
    """The metric's second half: |AGS(GPU pipeline) - AGS(reference)| on the reference's own inputs, through run_pipeline
    (native sampler -> HIP search -> classification -> estimate).  The reference values are the committed goldens produced by
    running the reference here (tests/golden/*.json); nothing of /root/reference is read."""
    import contextlib
    import io
    from microbecensus_amd import microbe_census as mc
    out = {}
    for case, seqfile, extra in (("config1_example_fq", "example.fq.gz", {"nreads": 10000, "read_length": 100}), ("unittest_metagenome", "metagenome.fa.gz", {})):
        gold = os.path.join(REPO, "tests", "golden", case + ".json")
        inp = os.path.join(REPO, "tests", "golden", "inputs", seqfile)
        if not (os.path.exists(gold) and os.path.exists(inp)):
            continue
        want = json.load(open(gold))["est_ags"]
        args = {"seqfiles": [inp], "device": device}
        args.update(extra)
        with contextlib.redirect_stdout(io.StringIO()):
            res = mc.run_pipeline(args)
        out[case] = None if res is None else abs(res[0] - want)
    return out


def m8_md5(lines):
    h = hashlib.md5()
    for ln in lines:
        h.update(ln.encode() if isinstance(ln, str) else ln)
    return h.hexdigest()


def cpu_baseline(eng, sample_reads, read_len, plan):
    """Times the reference's own engine (oracle/_ref: the bundled rapsearch binary + the canonical database) on the host cores on
    bounded prefixes of the bench reads, at several thread counts, and compares the md5 of its m8 body with the md5 of the m8 the
    GPU produced for the same reads.  plan: [(threads, nreads), ...]; the first entry is the headline value."""
    ref = os.path.join(REPO, "oracle", "_ref")
    rap, db = os.path.join(ref, "rapsearch_Linux_2.15"), os.path.join(ref, "rapdb_2.15")
    port = os.path.join(REPO, "oracle", "rs_port")
    kind = "reference" if (os.path.exists(rap) and os.path.exists(db)) else "port"
    if kind == "port" and not (os.path.exists(port) and os.path.exists(db)):
        return None
    nmax = max(n for _, n in plan)
    sample_reads = sample_reads[:nmax]
    runs = []
    with tempfile.TemporaryDirectory() as td:
        eng.upload(sample_reads); eng.run(0)
        gm8 = os.path.join(td, "gpu.m8")
        eng.write_m8(gm8)
        gpu_lines = open(gm8).readlines()
        gq = [int(l.split("\t", 1)[0]) for l in gpu_lines]
        for threads, n in plan:
            n = min(n, sample_reads.shape[0])
            fa = os.path.join(td, "sample_%d.fa" % n)
            if not os.path.exists(fa):
                with open(fa, "w") as f:
                    f.write("".join(">%d\n%s\n" % (i, bytes(r).decode()) for i, r in enumerate(sample_reads[:n])))
            out = os.path.join(td, "out_%d_%d" % (threads, n))
            if kind == "reference":
                cmd = [rap, "-q", fa, "-d", db, "-o", out, "-z", str(threads), "-e", "1", "-t", "n", "-p", "f", "-b", "0"]
                t = time.time(); subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL); dt = time.time() - t
                got = [l for l in open(out + ".m8") if not l.startswith("#")]
            else:
                t = time.time(); subprocess.check_call([port, db, fa, out + ".m8"]); dt = time.time() - t
                got = open(out + ".m8").readlines()
                threads = 1
            want = [l for l, q in zip(gpu_lines, gq) if q < n]
            runs.append({"threads": threads, "reads": n, "wall_s": round(dt, 2), "reads_per_s": round(n / dt, 1), "m8_rows": len(got),
                         "m8_md5_equals_gpu": m8_md5(got) == m8_md5(want)})
    head = runs[0]
    return {"value": head["reads_per_s"], "unit": "reads/s", "cores": head["threads"], "kind": kind,
            "sample": "prefixes of the bench workload (%d bp): %s -e 1 -t n -p f -b 0, wall time of the process incl. DB load; headline = first run" %
                      (read_len, "rapsearch_Linux_2.15 -z T" if kind == "reference" else "oracle/rs_port (C restatement, 1 thread)"),
            "host_cores": os.cpu_count(), "runs": runs, "m8_md5_equals_gpu": all(r["m8_md5_equals_gpu"] for r in runs)}


def e2e_rate(device, gen, n, L, gz):
    """File -> AGS through run_pipeline (native reader, HIP search, classification, estimate) on a FASTQ file of n reads of the
    bench workload written to the box's temp directory.  Returns reads/s of the whole call (file open to estimate)."""
    import contextlib
    import gzip
    import io
    import numpy as np
    from microbecensus_amd import microbe_census as mc
    reads = gen.single(n, L, first=1 << 40).cpu().numpy()      # (indices far away from the resident set)
    w = len(str(n - 1))
    rec = np.empty((n, 1 + w + 1 + L + 3 + L + 1), dtype=np.uint8)
    rec[:, 0] = ord("@")
    ids = np.arange(n)
    for k in range(w):
        rec[:, w - k] = ord("0") + (ids // 10 ** k) % 10
    rec[:, 1 + w] = 10
    rec[:, 2 + w:2 + w + L] = reads
    rec[:, 2 + w + L:5 + w + L] = np.frombuffer(b"\n+\n", dtype=np.uint8)
    rec[:, 5 + w + L:5 + w + 2 * L] = ord("I")
    rec[:, 5 + w + L + 7:5 + w + 2 * L:10] = ord("5")          # (a character only phred+33 files hold: the offset detection stops at the first record)
    rec[:, -1] = 10
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "reads.fq" + (".gz" if gz else ""))
        if gz:
            with gzip.open(path, "wb", compresslevel=1) as f:
                f.write(rec.tobytes())
        else:
            rec.tofile(path)
        size = os.path.getsize(path)
        walls = []
        for rep in range(2):                                       # the first call also allocates the engine's pools for this batch size
            args = {"seqfiles": [path], "device": device, "nreads": n, "read_length": L}
            t = time.time()
            with contextlib.redirect_stdout(io.StringIO()):
                res = mc.run_pipeline(args)
            walls.append(time.time() - t)
            if res is None:
                return None
    dt = walls[-1]
    return {"reads": n, "file": "FASTQ" + (".gz" if gz else ""), "file_bytes": size, "wall_s": round(dt, 3), "reads_per_s": round(n / dt, 1),
            "first_call_wall_s": round(walls[0], 3), "sampled_reads": int(res[1]["sampled_reads"]), "est_ags": res[0]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=2_000_000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--insert", type=int, default=300, help="fragment length of the paired library (N > 1)")
    ap.add_argument("--workload", choices=["genomes", "orfs"], default="genomes", help="genomes: reads of the 30 real genomes (SURVEY 8d); orfs: round 1's artificial ORF community")
    ap.add_argument("--resident-batches", type=int, default=0, help="distinct batches resident in HBM (0 = one per step: no read is searched twice in the timed region)")
    ap.add_argument("--cpu-sample", type=int, default=200_000, help="reads of the largest cpu_baseline run (-z all cores)")
    ap.add_argument("--cpu-full", action="store_true", help="cpu_baseline on >= 1 M reads at -z 1 / 8 / all cores (takes ~20 min)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ags-check", action="store_true", help="skip the run_pipeline AGS comparison on the reference's own inputs")
    ap.add_argument("--e2e-reads", type=int, default=20_000_000, help="reads of the end-to-end (file -> AGS) measurement (plain FASTQ; a tenth of it for .gz); 0 = skip")
    ap.add_argument("--count-in-timed-steps", action="store_true", help="keep the seed kernel's traffic counters on in the timed steps")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help="nccl = RCCL, one GPU per rank (the measurement); gloo: all ranks on GPU 0, reductions on the host - "
                                                                               "only to exercise the N > 1 code path on a one-GPU box")
    args = ap.parse_args()

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    if env_world is not None and int(env_world) != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%s" % (args.gpus, env_world))

    import numpy as np
    import torch
    import torch.distributed as dist
    from microbecensus_amd import _native, distributed as mcd, synth
    from microbecensus_amd import microbe_census as mc

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    if args.backend == "gloo":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    rdev = dev if args.backend == "nccl" else torch.device("cpu")   # where the tensors of the collectives live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)  # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend="gloo")

    names, seqs = _native.load_markers()
    model = _native.load_model()
    fams = model["families"]
    L = args.read_len
    eng = _native.Engine(device=local)
    eng.set_run(L, model["pars"][str(L)], fams)

    K = args.steps
    nres = args.resident_batches if args.resident_batches > 0 else max(1, K)
    per_rank = args.batch * nres
    gen = None
    if args.workload == "genomes":
        gen = synth.GenomeReads(device=dev, seed=20261001)
        if world == 1:
            reads = gen.single(per_rank, L)
            wl = "BASELINE configs[2]: error-free %d bp reads sampled from the reference's 30 genomes" % L
        else:
            # the paired library as the reference would consume it: file 1 (mate 1 of every fragment), then file 2
            total = per_rank * world
            F = total // 2
            lo, hi = rank * per_rank, (rank + 1) * per_rank
            parts = []
            if lo < F:
                parts.append(gen.paired(min(hi, F) - lo, L, frag=args.insert, first=lo)[0])
            if hi > F:
                a = max(lo, F) - F
                parts.append(gen.paired(hi - F - a, L, frag=args.insert, first=a)[1])
            reads = torch.cat(parts) if len(parts) > 1 else parts[0]
            wl = "BASELINE configs[3] (weak-scaling form): error-free paired %d bp reads (insert %d) of the 30 genomes, stream = file 1 then file 2, contiguous block per rank" % (L, args.insert)
    else:
        genome = synth.build_genomes(seqs, total_bp=8_000_000, seed=20261001)
        reads = torch.from_numpy(synth.sample_reads(genome, per_rank, L, seed=1000 + rank)).to(dev)
        wl = "artificial ORF community (round 1 workload), %d bp" % L
    torch.cuda.synchronize()
    eng.attach(reads.data_ptr(), per_rank)
    nf = len(fams)
    tot_hits = np.zeros(nf, np.int64); tot_aln = np.zeros(nf, np.int64); tot_bylen = np.zeros((nf, mcd.MAX_TARGET_LEN), np.int64)

    def step(i, collect=True):
        b = i % nres
        eng.run_range(b * args.batch, args.batch, first_read_id=b * args.batch)
        best = eng.best_hits(copy=False)    # the rows of the batch are in host memory too (mc_result_rows); the aggregation needs the best hits
        hits, aln, bylen = mcd.family_accumulators(best, nf)
        if world > 1:                       # RCCL: per-family hit counts / alignment sums of this step over all GPUs
            t = torch.from_numpy(np.concatenate([hits, aln])).to(rdev)
            dist.all_reduce(t)
            t = t.cpu().numpy()
            hits, aln = t[:nf], t[nf:]
        if collect:
            tot_hits.__iadd__(hits); tot_aln.__iadd__(aln); tot_bylen.__iadd__(bylen)
        return eng.stats()

    # The algorithmic traffic of the seed kernel (index reads of the reference's algorithm) depends on the reads only:
    # it is counted once per resident batch by untimed launches with the counters on, the timed steps run without them.
    eng.set_counting(True)
    traffic = []
    for b in range(nres):
        st = step(b, collect=False)
        traffic.append((st["bucket_lookups"], st["key_probes"]))
    eng.set_counting(bool(args.count_in_timed_steps))
    for i in range(args.warmup):
        step(i, collect=False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.time()
    acc = {}
    for i in range(K):
        st = step(i)
        st["bucket_lookups"], st["key_probes"] = traffic[i % nres]
        for k, v in st.items():
            acc[k] = acc.get(k, 0) + v
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.time() - t0
    # the same kernels one at a time (no overlap of the two parts): their own durations, outside the timed region
    eng.set_parts(1)
    seq = {}
    nseq = min(2, K)
    for i in range(nseq):
        b = i % nres
        eng.run_range(b * args.batch, args.batch, first_read_id=b * args.batch)
        for k, v in eng.stats().items():
            seq[k] = seq.get(k, 0) + v
    eng.set_parts(1)
    tmax = torch.tensor([dt], dtype=torch.float64, device=rdev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tb = torch.from_numpy(tot_bylen).to(rdev)
        dist.all_reduce(tb)                 # the 'cov' numerators (alignment length per target length): once, at the end
        tot_bylen = tb.cpu().numpy()
        cnt = torch.tensor([acc["rows"], acc["reads_with_rows"], acc["hsps"], acc["gap_tasks"], acc["seed_tasks"]], dtype=torch.int64, device=rdev)
        dist.all_reduce(cnt)
        job = dict(zip(("rows", "reads_with_rows", "hsps", "gap_tasks", "seed_tasks"), [int(x) for x in cnt.tolist()]))
    else:
        job = {k: int(acc[k]) for k in ("rows", "reads_with_rows", "hsps", "gap_tasks", "seed_tasks")}
    dt = float(tmax.item())

    if rank == 0:
        reads_total = args.batch * K * world
        # dominant kernel = the one with the largest accumulated HIP-event time (rank 0's events, on the library's own stream)
        kern = {"k_translate_seg": acc["ms_translate"], "k_enumerate": acc["ms_seed"], "k_eval_seeds": acc["ms_eval"], "k_gapped": acc["ms_gapped"],
                "sort": acc["ms_sort"], "k_finish": acc["ms_finish"]}
        SEQ = {"k_translate_seg": "ms_translate", "k_enumerate": "ms_seed", "k_eval_seeds": "ms_eval", "k_gapped": "ms_gapped", "sort": "ms_sort", "k_finish": "ms_finish"}
        kseq = {k: seq[m] / nseq for k, m in SEQ.items()}           # ms per step, every kernel alone on the GPU
        dom = max(kseq, key=kseq.get)
        n_batch = args.batch
        per_launch = {
            # algorithmic bytes per launch: what the reference's algorithm reads / writes for the same reads
            "k_translate_seg": n_batch * (L + 6 * (L // 3)),
            # the seed kernel's OWN algorithm: frames in, one bucket-bitmap word per seed position, what it asks its filters (9-mer
            # filter word 4 B, wildcard line 32 B, pair block 16 B), bucket record + key group per surviving probe (32 + 16 B), posting
            # in (4 B) and seed hit out (16 B) per hit - counted by the timed kernel itself (mc_stats.seed_*)
            "k_enumerate": (n_batch * (6 * (L // 3) + 4 * seed_positions(L)) * K + 4 * acc["seed_exact_asks"] + 32 * acc["seed_wild_asks"] + 16 * acc["seed_pair_asks"]
                            + 48 * acc["seed_probes"] + 20 * acc["seed_tasks"]) / K,
            "k_eval_seeds": (acc["seed_tasks"] * (16 + 4 + 8 + 2 * 20)) / K,
            "k_gapped": (acc["gap_tasks"] * 24 + acc["hsps"] * 48) / K,
            "sort": acc["hsps"] * (12 * 4 + 48 * 2) / K,
            "k_finish": acc["hsps"] * 48 * 3 / K,
        }
        # the index reads the REFERENCE's algorithm would issue for the same reads (counting form of the seed kernel): what the filters dispose of
        ref_seed_bytes = (n_batch * 6 * (L // 3) * K + 8 * acc["bucket_lookups"] + 2 * acc["key_probes"] + 20 * acc["seed_tasks"]) / K
        # The two parts of a step overlap in the timed region, so a kernel's HIP events there also span the other part's kernels;
        # its own duration is measured by the same events right after the timed region with one kernel at a time (mc_set_parts(1)),
        # which is also how the committed rocprofv3 profile (profiles/, MC_PARTS=1) is taken.  achieved / frac use that duration;
        # achieved_in_timed_region uses the overlapped one.
        ach = per_launch[dom] / (kseq[dom] * 1e-3) / 1e9
        ach_timed = per_launch[dom] / (kern[dom] / K * 1e-3) / 1e9
        traffic_dom = profiled_traffic("k_enumerate_t0" if dom == "k_enumerate" else dom, n_batch)
        # the whole device pipeline of one batch as one "launch": SURVEY 8(d)'s own definition, achieved = reads/s x A(L)
        pipe_gbs = (SURVEY_A[L] * (reads_total / world) / dt / 1e9) if L in SURVEY_A else None
        agg = mcd.aggregate_from_accumulators(tot_hits, tot_aln, tot_bylen, fams, mc.find_opt_pars(None, L))
        try:
            est = mc.estimate_average_genome_size({"read_length": L, "sampled_reads": reads_total, "verbose": False}, None, agg)
        except BaseException:
            est = None
        out = {
            "metric": METRIC, "value": round(reads_total / dt, 1), "unit": "reads/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": round(dt / K * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/int32",
            "data": "synthetic",
            "config": {"workload": "%s; %d reads/step/GPU, %d steps, %d distinct reads resident in HBM per GPU" % (wl, n_batch, K, per_rank),
                       "read_len": L, "batch": n_batch, "parallelism": "reads sharded over %d GPU(s), %s all_reduce of per-family accumulators per step" % (world, "RCCL" if args.backend == "nccl" else "gloo (plumbing test: all ranks on one GPU)"),
                       "marker_db": "%d proteins / %d families" % (len(names), len(fams)),
                       "classified_reads": int(tot_hits.sum()), "classified_per_read": round(float(tot_hits.sum()) / reads_total, 6),
                       "rows_per_read": round(job["rows"] / reads_total, 4), "reads_with_rows": round(job["reads_with_rows"] / reads_total, 5),
                       "hsps_per_read": round(job["hsps"] / reads_total, 3), "gapped_extensions_per_read": round(job["gap_tasks"] / reads_total, 3),
                       "seed_hits_per_read": round(job["seed_tasks"] / reads_total, 2), "ags_estimate_of_workload": est,
                       # HIP events on each part's own stream inside the timed region: the two parts of a step overlap, so these
                       # durations include waiting for the other part's kernels (their sum exceeds ms_per_step)
                       "kernel_ms_per_step": {k: round(v / K, 3) for k, v in kern.items()},
                       "sum_kernel_ms_per_step": round(sum(kern.values()) / K, 3),
                       # the same kernels run one at a time after the timed region (mc_set_parts(1))
                       "kernel_ms_per_step_sequential": {k: round(seq[m] / nseq, 3) for k, m in (("k_translate_seg", "ms_translate"), ("k_enumerate", "ms_seed"),
                                                         ("k_eval_seeds", "ms_eval"), ("k_gapped", "ms_gapped"), ("sort", "ms_sort"), ("k_finish", "ms_finish"))}},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
                         "traffic": (None if traffic_dom is None else round(traffic_dom, 0)),
                         "basis": "achieved = algorithmic bytes of one step's launches of the kernel (DESIGN.md 5; for the seed kernel: what its own algorithm "
                                  "asks for - frames, bitmap words, filter words / lines / blocks, bucket records, key groups, postings, seed hits - counted by the "
                                  "timed kernel) / HIP-event time of those launches, one kernel at a time (measured right after the timed region; in it the two "
                                  "parts of a step overlap). traffic / physical_* = fabric-side bytes (2 x FETCH_SIZE + WRITE_SIZE) of the committed rocprofv3 "
                                  "PMC profile: above the algorithmic bytes because a 16- or 32-byte item arrives as a 128-byte line; mostly Infinity-Cache "
                                  "hits (the index is 110 MB). seed_kernel_reference_algorithm_*: the index reads the reference's algorithm would issue for the "
                                  "same reads / the same time - a disposal rate, the filters answer those probes",
                         "seed_kernel_reference_algorithm_bytes_per_read": round(ref_seed_bytes / n_batch, 1),
                         "seed_kernel_reference_algorithm_GBps": round(ref_seed_bytes / (kseq["k_enumerate"] * 1e-3) / 1e9, 2),
                         "physical_GBps": (None if traffic_dom is None else round(traffic_dom / (kseq[dom] * 1e-3) / 1e9, 2)),
                         "physical_frac": (None if traffic_dom is None else round(traffic_dom / (kseq[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)),
                         "kernel_ms_per_step": round(kseq[dom], 3),
                         "l2_hit_rate": profiled_l2_hit_rate("k_enumerate_t0" if dom == "k_enumerate" else dom),
                         "achieved_in_timed_region": round(ach_timed, 2),
                         "algorithmic_bytes_per_read": round(per_launch[dom] / n_batch, 1),
                         # SURVEY.md 8(d) priced the whole path at A(150) = 207,923 B/read assuming whole-bucket visits the engine does not
                         # perform; its definition achieved = reads/s x A(L), per GPU:
                         "survey_A_bytes_per_read": SURVEY_A.get(L),
                         "pipeline_GBps_with_survey_A": (None if pipe_gbs is None else round(pipe_gbs, 2)),
                         "pipeline_frac_with_survey_A": (None if pipe_gbs is None else round(pipe_gbs / HBM_PEAK_GBS, 5)),
                         "all_kernels_algorithmic_GBps": {k: round(per_launch[k] / (kseq[k] * 1e-3) / 1e9, 2) for k in kseq if kseq[k] > 0},
                         "all_kernels_physical_GBps": {k: (lambda t: None if t is None else round(t / (kseq[k] * 1e-3) / 1e9, 2))(
                             profiled_traffic("k_enumerate_t0" if k == "k_enumerate" else k, n_batch)) for k in kseq if kseq[k] > 0 and k != "sort"}},
        }
        if world == 1 and not args.no_cpu_baseline:
            cores = os.cpu_count() or 1
            if args.cpu_full:
                plan = [(cores, 1_000_000), (8, 1_000_000), (1, 1_000_000)]
            else:   # bounded: about 10-15 s of wall time per run
                plan = [(cores, args.cpu_sample), (8, max(1000, args.cpu_sample // 5)), (1, max(1000, args.cpu_sample // 15))]
            ns = min(max(n for _, n in plan), per_rank)
            out["cpu_baseline"] = cpu_baseline(eng, reads[:ns].cpu().numpy(), L, [(t, min(n, ns)) for t, n in plan])
        if world == 1 and not args.no_ags_check:
            out["config"]["ags_abs_error_vs_reference"] = ags_abs_error(local)
        if world == 1 and args.e2e_reads > 0 and gen is not None:
            out["e2e"] = {"what": "run_pipeline(file -> AGS): native reader beside the HIP search (mc_search_files), classification, estimate; wall time of the second "
                                  "of two calls on the same file (first_call_wall_s includes the one-time pool allocation); .gz is bounded by single-stream inflate",
                          "plain": e2e_rate(local, gen, args.e2e_reads, L, gz=False), "gz": e2e_rate(local, gen, max(1, args.e2e_reads // 10), L, gz=True)}
        print(json.dumps(out))
        sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
