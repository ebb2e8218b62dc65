#!/usr/bin/env python3
"""bench.py - reads/s of the MI355X translated-search hot path on synthetic 150 bp reads.

    python bench.py --gpus N --steps K --warmup W
(for N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[2], the 150 bp single-GPU configuration the metric is quoted on):
synthetic error-free 150 bp shotgun reads, resident in HBM before the timed region starts.  One "step" is
one pass of the whole device pipeline (translate+SEG, seeds, extension, ranking, classification) over one
batch of --batch reads; step i works on batch i mod (resident batches).  With the defaults, K*batch = 20 M
reads per GPU.  Reads shard across ranks with no data-path collective; per-family hit counts are summed
with one RCCL all_reduce per step (weak scaling: every rank processes its own K batches).

Rank 0 prints ONE JSON line: metric/value (whole-job reads/s), roofline of the dominant kernel
(algorithmic bytes counted by the kernel itself / HIP-event duration), and - at N=1 - the reference's own
RAPsearch2 binary timed on this host's cores on a bounded sample of the same reads (cpu_baseline).
"""
import argparse
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
SURVEY_A = {100: 127403, 150: 207923, 300: 452893}   # modelled algorithmic bytes per read of the seed path


def torch_splitmix64(x):
    import torch
    M = lambda v: torch.tensor(v, dtype=torch.int64, device=x.device)  # noqa: E731
    lsr = lambda v, k: (v >> k) & ((1 << (64 - k)) - 1)                # noqa: E731  logical shift on int64
    x = x + M(-7046029254386353131)         # 0x9E3779B97F4A7C15
    z = x
    z = (z ^ lsr(z, 30)) * M(-4658895280553007687)   # 0xBF58476D1CE4E5B9
    z = (z ^ lsr(z, 27)) * M(-7723592293110705685)   # 0x94D049BB133111EB
    return z ^ lsr(z, 31)


def sample_reads_device(genome_np, nreads, read_len, seed, device):
    """Same reads as microbecensus_amd.synth.sample_reads, generated directly in HBM with torch."""
    import numpy as np
    import torch
    from microbecensus_amd.synth import splitmix64
    g = torch.from_numpy(genome_np).to(device)
    rc = torch.zeros(256, dtype=torch.uint8, device=device)
    for a, b in zip(b"ACGTN", b"TGCAN"):
        rc[a] = b
    base = int(splitmix64(np.array([seed], dtype=np.uint64))[0])
    base = base - (1 << 64) if base >= (1 << 63) else base
    mul = -3372029247567499371 & 0xFFFFFFFFFFFFFFFF   # 0xD1342543DE82EF95
    mul = mul - (1 << 64) if mul >= (1 << 63) else mul
    out = torch.empty((nreads, read_len), dtype=torch.uint8, device=device)
    span = len(genome_np) - read_len
    ar = torch.arange(read_len, device=device, dtype=torch.int64)
    chunk, k = 1 << 20, 0
    for s in range(0, nreads, chunk):
        m = min(chunk, nreads - s)
        idx = torch.arange(k, k + m, device=device, dtype=torch.int64)
        v1 = torch_splitmix64(idx * mul + base); k += m
        idx = torch.arange(k, k + m, device=device, dtype=torch.int64)
        v2 = torch_splitmix64(idx * mul + base); k += m
        # unsigned v1 % span
        start = (((v1 >> 1) & 0x7FFFFFFFFFFFFFFF) % span * 2 + (v1 & 1)) % span
        rev = (v2 & 1).bool()
        block = g[start[:, None] + ar[None, :]]
        rblock = rc[torch.flip(block, dims=[1]).long()]
        out[s:s + m] = torch.where(rev[:, None], rblock, block)
    return out


def profiled_traffic(kernel, n_batch):
    """HBM-side bytes per launch of `kernel` from the newest committed rocprofv3 PMC summary (profiles/rNN_hbm_traffic.json:
    FETCH_SIZE and WRITE_SIZE collected in their own --pmc passes; read bytes = 2 x FETCH_SIZE on gfx950, see
    profiles/README.md), scaled from the profiled batch to this run's batch.  None if no profile is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_hbm_traffic.json")))
    if not files:
        return None
    try:
        tr = json.load(open(files[-1]))
        line = json.load(open(files[-1].replace("_hbm_traffic.json", "_bench_line.json")))
        per = float(line["config"]["batch"])
        key = [k for k in tr if k.startswith(kernel) and "true" not in k]
        if not key or tr[key[0]].get("write_kib_per_launch") is None:
            return None
        t = tr[key[0]]
        return (2.0 * t["fetch_kib_per_launch"] + t["write_kib_per_launch"]) * 1024.0 / per * n_batch
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


def cpu_baseline(sample_reads, read_len, eng_rows):
    """Times the reference's own engine on the host cores on a bounded sample of the bench reads."""
    import numpy as np
    ref = os.path.join(REPO, "oracle", "_ref")
    cores = os.cpu_count() or 1
    n = sample_reads.shape[0]
    with tempfile.TemporaryDirectory() as td:
        fa = os.path.join(td, "sample.fa")
        with open(fa, "w") as f:
            f.write("".join(">%d\n%s\n" % (i, bytes(r).decode()) for i, r in enumerate(sample_reads)))
        rap, db = os.path.join(ref, "rapsearch_Linux_2.15"), os.path.join(ref, "rapdb_2.15")
        if os.path.exists(rap) and os.path.exists(db):
            kind = "reference"
            cmd = [rap, "-q", fa, "-d", db, "-o", os.path.join(td, "out"), "-z", str(cores), "-e", "1", "-t", "n", "-p", "f", "-b", "0"]
            t = time.time(); subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL); dt = time.time() - t
            got = [l for l in open(os.path.join(td, "out.m8")) if not l.startswith("#")]
        else:
            kind = "port"
            port = os.path.join(REPO, "oracle", "rs_port")
            if not (os.path.exists(port) and os.path.exists(db)):
                return None
            t = time.time(); subprocess.check_call([port, db, fa, os.path.join(td, "out.m8")]); dt = time.time() - t
            got = open(os.path.join(td, "out.m8")).readlines()
    same = None
    if eng_rows is not None:
        same = (len(got) == len(eng_rows))
    return {"value": round(n / dt, 1), "unit": "reads/s", "cores": cores, "kind": kind,
            "sample": "first %d reads of the bench workload (%d bp), rapsearch -z %d -e 1 -t n -p f -b 0, wall %.1f s incl. DB load" % (n, read_len, cores, dt),
            "rows_match_gpu": same}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=2_000_000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--resident-batches", type=int, default=4)
    ap.add_argument("--cpu-sample", type=int, default=200_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ags-check", action="store_true", help="skip the run_pipeline AGS comparison on the reference's own inputs")
    ap.add_argument("--count-in-timed-steps", action="store_true", help="keep the seed kernel's traffic counters on in the timed steps")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from microbecensus_amd import _native, synth

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)

    names, seqs = _native.load_markers()
    model = _native.load_model()
    fams = model["families"]
    L = args.read_len
    eng = _native.Engine(device=local)
    eng.set_run(L, model["pars"][str(L)], fams)

    genome = synth.build_genomes(seqs, total_bp=8_000_000, seed=20261001)
    nres = min(args.resident_batches, max(1, args.steps))
    total_resident = args.batch * nres
    reads = sample_reads_device(genome, total_resident, L, seed=1000 + rank, device=dev)
    torch.cuda.synchronize()
    eng.attach(reads.data_ptr(), total_resident)
    fam_counts = torch.zeros(len(fams), dtype=torch.int64, device=dev)

    def step(i):
        b = i % nres
        eng.run_range(b * args.batch, args.batch, first_read_id=b * args.batch)
        best = eng.best_hits(copy=False)    # the rows of the batch are in host memory too (mc_result_rows); the aggregation needs the best hits
        c = np.bincount(best["family"], minlength=len(fams)).astype(np.int64)
        t = torch.from_numpy(c).to(dev)
        if world > 1:
            dist.all_reduce(t)          # RCCL: per-family hit counts of this step over all GPUs
        fam_counts.add_(t)
        return eng.stats()

    # The algorithmic traffic of the seed kernel (index reads of the reference's algorithm) depends on the reads only:
    # it is counted once per resident batch by untimed launches with the counters on, the timed steps run without them.
    eng.set_counting(True)
    traffic = []
    for b in range(nres):
        st = step(b)
        traffic.append((st["bucket_lookups"], st["key_probes"]))
    eng.set_counting(bool(args.count_in_timed_steps))
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.time()
    acc = {}
    for i in range(args.steps):
        st = step(args.warmup + i)
        st["bucket_lookups"], st["key_probes"] = traffic[(args.warmup + i) % nres]
        for k, v in st.items():
            acc[k] = acc.get(k, 0) + v
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.time() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    if rank == 0:
        K = args.steps
        reads_total = args.batch * K * world
        # dominant kernel = the one with the largest accumulated HIP-event time
        kern = {"k_translate_seg": acc["ms_translate"], "k_enumerate": acc["ms_seed"], "k_eval_seeds": acc["ms_eval"], "k_gapped": acc["ms_gapped"],
                "sort": acc["ms_sort"], "k_finish": acc["ms_finish"]}
        dom = max(kern, key=kern.get)
        n_batch = args.batch
        per_launch = {
            # algorithmic bytes per launch, counted by the kernels themselves (DESIGN.md "Measurement")
            "k_translate_seg": n_batch * (L + 6 * (L // 3)),
            "k_enumerate": (n_batch * 6 * (L // 3) * K + 8 * acc["bucket_lookups"] + 2 * acc["key_probes"] + 20 * acc["seed_tasks"]) / K,
            "k_eval_seeds": (acc["seed_tasks"] * (16 + 4 + 8 + 2 * 20)) / K,
            "k_gapped": (acc["gap_tasks"] * 24 + acc["hsps"] * 48) / K,
            "sort": acc["hsps"] * (12 * 4 + 48 * 2) / K,
            "k_finish": acc["hsps"] * 48 * 3 / K,
        }
        ach = per_launch[dom] / (kern[dom] / K * 1e-3) / 1e9
        out = {
            "metric": METRIC, "value": round(reads_total / dt, 1), "unit": "reads/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": round(dt / K * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/int32",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: synthetic error-free %d bp reads, %d reads/step/GPU, %d steps (=%d reads/GPU), resident in HBM" % (L, n_batch, K, n_batch * K),
                       "read_len": L, "batch": n_batch, "parallelism": "reads sharded over %d GPU(s), RCCL all_reduce of per-family hit counts" % world,
                       "marker_db": "%d proteins / %d families" % (len(names), len(fams)),
                       "classified_reads": int(fam_counts.sum().item()),
                       "kernel_ms_per_step": {k: round(v / K, 3) for k, v in kern.items()}},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
                         "traffic": (lambda t: None if t is None else round(t, 0))(profiled_traffic("k_enumerate_t0" if dom == "k_enumerate" else dom, n_batch)),
                         "algorithmic_bytes_per_read": round(per_launch[dom] / n_batch, 1),
                         # SURVEY.md 8(d) priced the seed path at A(150) = 207,923 B/read assuming whole-bucket visits the engine does not
                         # perform; for comparison, the same kernel time priced with that figure:
                         "survey_A_bytes_per_read": SURVEY_A.get(L), "seed_kernel_GBps_with_survey_A": (round(SURVEY_A[L] * n_batch / (kern["k_enumerate"] / K * 1e-3) / 1e9, 2) if L in SURVEY_A and kern["k_enumerate"] > 0 else None),
                         "all_kernels_GBps": {k: round(per_launch[k] / (kern[k] / K * 1e-3) / 1e9, 2) for k in kern if kern[k] > 0}},
        }
        if world == 1 and not args.no_cpu_baseline:
            ns = min(args.cpu_sample, args.batch)
            sample = reads[:ns].cpu().numpy()
            eng.upload(sample); eng.run(0)
            rows, _ = eng.results()
            out["cpu_baseline"] = cpu_baseline(sample, L, rows)
        if world == 1 and not args.no_ags_check:
            out["config"]["ags_abs_error_vs_reference"] = ags_abs_error(local)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
