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
the per-family accumulators of every step are summed over the ranks with an RCCL all_reduce.  The steps are issued the way
mc_search / mc_search_files issue their batches: mc_range_end(i), mc_range_begin(i + 1), then the results of step i - the front
of the next step is enqueued before the host sums up (--one-at-a-time: mc_run_range per step).  The timed region holds exactly
K steps, each from its first kernel to its results on the host.

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
import microbecensus_amd  # noqa: E402
microbecensus_amd.configure_process_env()     # bench.py owns its process (and its ranks inherit the environment): GPU_MAX_HW_QUEUES=8 unless set

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
    "k_translate_seg": ["k_translate_seg"], "k_enumerate_t0": ["k_enumerate_q<", "k_enumerate_t0<"], "k_enumerate": ["k_enumerate_q<", "k_enumerate_t0<", "k_enumerate"], "k_eval_seeds": ["k_eval_seeds"],
    "k_gapped": ["k_gap_dedupe", "k_gap_sort_hist", "k_gap_sort_scan", "k_gap_sort_scatter", "k_gapped_lds", "k_gapped", "k_gap_emit"], "k_finish": ["k_finish", "k_finish_heavy", "k_heap_lanes", "k_heavy_rows", "k_heavy_lists", "k_emit_rows"],
    "sort": ["k_bin_count", "k_bin_scatter", "k_scan_sums", "k_scan_top", "k_scan_apply", "k_order_lists", "k_order_light", "k_order_heavy", "k_order_copy"],
}


FETCH_FACTOR_STREAM = 2.0   # MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads
GATHER_KERNELS = ("k_enumerate_q<", "k_enumerate_t0<", "k_eval_seeds")   # kernels whose global reads are scattered 4..32-byte items


def load_profile(L):
    """The newest committed rocprofv3 summary taken AT THIS READ LENGTH (profiles/rNN_L<L>_*; tools/profile_round.sh): per-kernel
    durations (kernel-trace pass) and PMC counters per launch (their own --pmc passes).  None when no profile of this length is
    committed - a profile of another length is never scaled."""
    import csv
    import glob
    stats = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_L%d_kernel_stats.csv" % L)))
    if not stats:
        return None
    base = stats[-1][: -len("_kernel_stats.csv")]
    prof = {"files": os.path.basename(base) + "_{kernel_stats.csv,pmc_per_launch.csv,hbm_traffic.json,bench_line.json}", "kernels": {}, "reads_per_launch": None}
    try:
        prof["reads_per_launch"] = int(json.load(open(base + "_bench_line.json"))["config"]["batch"])
        for r in csv.DictReader(open(base + "_kernel_stats.csv")):
            prof["kernels"].setdefault(r["Name"], {})["avg_ns"] = float(r["AverageNs"])
            prof["kernels"][r["Name"]]["calls"] = int(r["Calls"])
        for r in csv.DictReader(open(base + "_pmc_per_launch.csv")):
            d = prof["kernels"].setdefault(r["Kernel"], {})
            for k, v in r.items():
                if k not in ("Kernel", "Launches") and v not in ("", None):
                    d[k] = float(v)
        cal = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_fetch_calibration.json")))
        prof["gather_factor"], prof["calibration"] = None, None
        if cal:
            c = json.load(open(cal[-1]))
            g = c.get("k_gather<32>", {})
            if g.get("FETCH_SIZE_over_requested"):
                # bytes the fabric moved per byte FETCH_SIZE reports, for scattered 32-byte items: one 128-byte line per item
                prof["gather_factor"] = c.get("gather_fetch_factor")
                prof["calibration"] = os.path.basename(cal[-1])
    except Exception as e:                                    # a damaged profile is no profile
        sys.stderr.write("bench.py: profile %s unreadable: %s\n" % (base, e))
        return None
    return prof


def gather_ceiling():
    """The newest committed calibration of the machine's SCATTERED-LINE rate (profiles/rNN_gather_ceiling.json, made by
    tools/gather_ceiling.sh on the GPU box): G lines/s by footprint and item width, at 16 waves per CU (the figures do not move with
    8 or 32 waves per CU, with one or eight asks in flight per lane, or when every ask depends on the one before).  None when absent."""
    import glob
    cal = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_gather_ceiling.json")))
    if not cal:
        return None
    try:
        rows = json.load(open(cal[-1]))["rows"]
        rate = {(r["footprint"], r["bytes"]): r["g_lines_per_s"] for r in rows if r["kind"] == "gather" and r["waves_per_cu"] == 16}
        mix = [r["g_lines_per_s"] for r in rows if r["kind"] == "seed_mix_cascade" and r["waves_per_cu"] == 16]
        return {"file": os.path.basename(cal[-1]), "rate": rate, "mix": (mix[0] if mix else None)}
    except Exception as e:
        sys.stderr.write("bench.py: %s unreadable: %s\n" % (cal[-1], e))
        return None


def seed_line_model(cal, L, n_batch, asks, hits, seed_ms):
    """What the seed kernel's own asks cost at the calibrated scattered-line rates, structure by structure: every ask
    is one aligned item = one line asked of the level its structure lives in.  items per read x 1 / rate(footprint class, item width),
    summed, against the kernel's measured duration.  The structures' sizes are those of the marker database (118 MB of index).
    (A hit's posting and its subject's offsets - 150 of the lines a read of 150 bp cost the kernel - are no longer asked here: the hit
    record holds the posting's index and k_eval_seeds fetches them - with the subject's residues, in one 32-byte record: eval_line_model.)"""
    if cal is None or seed_ms <= 0:
        return None
    positions = sum(max(0, (L - f) // 3 - 6) for f in (0, 1, 2)) * 2          # seed positions of the six frames: each asks the bucket bitmap once
    per_read = [   # (structure, footprint class of the calibration, item bytes, asks per read)
        ("bucket bitmap 125 KB", "128KB", 4, float(positions)),
        ("9-mer filter 1 MB", "1MB", 4, asks["seed_exact_asks"] / n_batch),
        ("wildcard filter 16 MB", "16MB", 32, asks["seed_wild_asks"] / n_batch),
        ("pair filter 16 MB", "16MB", 16, asks["seed_pair_asks"] / n_batch),
        ("bucket records 32 MB", "32MB", 16, asks["seed_probes"] / n_batch),
        ("key groups 7 - 11 MB / range table (44 % of the probes: the others find an empty group)", "8MB", 16, 0.44 * asks["seed_probes"] / n_batch),
    ]
    rows, t = [], 0.0
    for name, foot, width, n in per_read:
        r = cal["rate"].get((foot, width))
        if not r:
            return None
        ms = n * n_batch / (r * 1e9) * 1e3
        t += ms
        rows.append({"structure": name, "asks_per_read": round(n, 1), "g_lines_per_s_at_that_footprint": r, "ms_per_launch": round(ms, 3)})
    lines = sum(x[3] for x in per_read)
    return {"calibration": cal["file"], "lines_asked_per_read": round(lines, 1), "achieved_g_lines_per_s": round(lines * n_batch / (seed_ms * 1e-3) / 1e9, 1),
            "ceiling_of_the_seed_mix_g_lines_per_s": cal["mix"], "model_ms_per_launch": round(t, 3), "measured_ms_per_launch": round(seed_ms, 3),
            "frac": round(t / seed_ms, 3), "by_structure": rows,
            "note": "frac = the time the machine needs for these asks at its measured scattered-line rates (each structure alone in the caches: the optimistic case) / the "
                    "seed kernel's measured time. About 1 means the kernel runs at the rate the memory system delivers scattered lines at - fewer or more local lines are "
                    "the only way down, not more lanes, waves or asks in flight (the calibration's rates are the same at 8, 16 and 32 waves per CU, with 1 or 8 asks in "
                    "flight per lane, and for a chain of dependent asks); above 1: asks of one wave that fall into the same line are counted once each"}


def eval_line_model(cal, n_batch, hits, survivors, eval_ms, prof_der):
    """The same pricing for k_eval_seeds: what a seed hit makes the kernel ask of structures that do not fit an L2 - its
    posting's record (posting, the subject's place and the 24 residues of the subject around the seed: one aligned 32-byte item of a
    115 MB array; until late in round 6 an 8-byte posting entry AND 24 bytes at a scattered place of the 14 MB residue array, 2.4 lines)
    and, for the hits that pass the gate, the residue array in front of and behind the grown seed (the two X-drop walks: a line each) -
    at the calibrated scattered-line rates, against the kernel's measured duration.  (Its records, the frames' rows and its outputs are
    streams.)  survivors: HSPs + gap tasks per launch - a lower bound (an ungapped HSP below the thresholds leaves no record)."""
    if cal is None or eval_ms <= 0 or not hits:
        return None
    r_rec, r_res = cal["rate"].get(("128MB", 32)), cal["rate"].get(("16MB", 32))
    if not r_rec or not r_res:
        return None
    per_hit = [("posting records 115 MB (32-byte items: posting, place, 24 residues of the subject)", 1.0, r_rec),
               ("subject residues 14 MB (the X-drop walks of the hits that pass the gate: two lines each)", 2.0 * survivors / hits, r_res)]
    rows, t = [], 0.0
    for name, n, r in per_hit:
        ms = n * hits / (r * 1e9) * 1e3
        t += ms
        rows.append({"structure": name, "lines_per_hit": round(n, 2), "g_lines_per_s_at_that_footprint": r, "ms_per_launch": round(ms, 3)})
    return {"calibration": cal["file"], "hits_per_read": round(hits / n_batch, 2), "model_ms_per_launch": round(t, 3), "measured_ms_per_launch": round(eval_ms, 3), "frac": round(t / eval_ms, 3),
            "l2_requests_per_hit_of_the_committed_profile": prof_der, "by_structure": rows,
            "note": "frac near 1: the ungapped extension kernel too takes the time the memory system needs for its scattered lines - halving the instructions of its X-drop "
                    "loops, 4 to 8 waves per SIMD and loads a turn ahead all left its duration where it was, moving the subject's residues INTO the posting's record (2.4 -> 1 "
                    "line per hit) took 11 - 15 % off (DESIGN.md 5.8); above 1: hits of one probe have neighbouring records and share lines"}


def stage_counters(prof, stage):
    """Sums the per-launch counters of the kernels of a stage (one launch of each per pass of the pipeline over the profiled batch;
    a kernel launched several times per pass counts with its number of calls per pass)."""
    if prof is None:
        return None
    pref = STAGE_KERNELS.get(stage, [stage])
    tot, names, per_pass = {}, [], None
    for k, d in prof["kernels"].items():
        if (k.rstrip().endswith("true>") and k.startswith("k_enumerate")) or k.startswith("k_enumerate_count"):       # (the counting form of the seed kernel is not the timed one)
            continue
        if not any(k == q or k.startswith(q + "<") or (q.endswith("<") and k.startswith(q)) for q in pref):
            continue
        names.append(k)
        calls = d.get("calls", 0)
        if per_pass is None and stage != "k_finish":
            per_pass = calls
        mult = 1.0
        if k == "k_finish":                                     # (profiles taken before the size classes shared one launch: four launches per pass)
            passes = prof["kernels"].get("k_heavy_lists", {}).get("calls", 0)
            mult = float(calls) / passes if passes else 1.0
        for c, v in d.items():
            if c in ("calls",):
                continue
            tot[c] = tot.get(c, 0.0) + v * mult
    if not names:
        return None
    tot["_kernels"] = sorted(names)
    return tot


def derived(prof, stage):
    """Counter-derived figures of a stage at the profiled batch size (every one recomputes from the committed files):
    issue_frac = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x kernel cycles) - a wave64 VALU instruction holds its SIMD for 4 cycles;
    salu_frac = SQ_INSTS_SALU / (256 scalar units x kernel cycles) - one scalar issue per CU and cycle (the two run side by side, so
    they are NOT added); kernel cycles = duration x 2.4 GHz (the peak clock: the fractions are lower bounds); wait_frac = SQ_WAIT_ANY /
    SQ_WAVE_CYCLES; lanes = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU (of 64); issue_roofline = issue_frac x lanes / 64 - the share
    of the machine's lane-cycles that did work: the fraction of the resource that actually binds these kernels; fabric_bytes = factor x
    FETCH_SIZE + WRITE_SIZE (KiB as rocprofv3 reports them; factor 2 for streaming reads, the gather calibration's upper bound for
    the two gather kernels) - what crossed the fabric behind the L2s (L2 misses: Infinity-Cache hits included, so NOT all of it
    is HBM traffic - the index is 110 MB; rounds 1 - 4 called this hbm_frac), fabric_frac = fabric_bytes / duration / 8 TB/s;
    l2_hit = TCC_HIT / (HIT + MISS)."""
    c = stage_counters(prof, stage)
    if not c or not c.get("avg_ns"):
        return None
    ns = c["avg_ns"]
    cycles = ns * 2.4
    out = {"kernels": c["_kernels"], "profiled_ms": round(ns / 1e6, 4), "reads_per_launch": prof["reads_per_launch"]}
    if c.get("SQ_INSTS_VALU"):
        out["issue_frac"] = round(c["SQ_INSTS_VALU"] * 4.0 / (1024.0 * cycles), 4)
        out["salu_frac"] = round(c.get("SQ_INSTS_SALU", 0.0) / (256.0 * cycles), 4)
    if c.get("SQ_WAVE_CYCLES"):
        out["wait_frac"] = round(c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 4)
    if c.get("SQ_ACTIVE_INST_VALU") and c.get("SQ_THREAD_CYCLES_VALU"):
        out["valu_lanes_of_64"] = round(c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"], 2)
        if "issue_frac" in out:
            out["issue_roofline"] = round(out["issue_frac"] * out["valu_lanes_of_64"] / 64.0, 4)
    if len(c["_kernels"]) == 1 and c.get("SQ_WAVES") and c["SQ_WAVES"] <= 256 * 32 and c.get("GRBM_GUI_ACTIVE") and c.get("SQ_WAVE_CYCLES"):   # (a kernel whose waves are all resident from the start)
        # a persistent kernel's waves should all live as long as the launch: the share of the launch a wave is resident
        # (SQ_WAVE_CYCLES counts in units of 4 cycles, GRBM_GUI_ACTIVE is summed over the 8 XCDs); what is missing is the tail of the
        # launch - the waves that got the cheap reads waiting for the one that got the expensive ones
        out["wave_residency"] = round(c["SQ_WAVE_CYCLES"] * 4.0 / (c["SQ_WAVES"] * c["GRBM_GUI_ACTIVE"] / 8.0), 3)
    if c.get("SQ_LDS_IDX_ACTIVE"):
        out["lds_bank_conflict_frac"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 4)
    if c.get("FETCH_SIZE") is not None and c.get("WRITE_SIZE") is not None:
        gather = any(k.startswith(GATHER_KERNELS) for k in c["_kernels"])
        f = (prof.get("gather_factor") or FETCH_FACTOR_STREAM) if gather else FETCH_FACTOR_STREAM
        hb = (f * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
        out["fetch_factor"] = f
        out["fabric_bytes_per_launch"] = round(hb, 0)
        out["fabric_frac"] = round(hb / (ns * 1e-9) / (HBM_PEAK_GBS * 1e9), 5)
        if gather:                                              # (a scattered request moves 64 or 128 bytes: lower bound with factor 1)
            out["fabric_frac_lower"] = round((1.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 / (ns * 1e-9) / (HBM_PEAK_GBS * 1e9), 5)
    if c.get("TCC_HIT_sum") is not None and (c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0)) > 0:
        out["l2_hit_rate"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
    # what binds the kernel, by the counters
    # (for the gather kernels the bytes that certainly moved - the lower bound - decide: the upper bound assumes a whole 128-byte line per item)
    cand = {"valu_issue": out.get("issue_frac", 0.0), "salu_issue": out.get("salu_frac", 0.0), "fabric": out.get("fabric_frac_lower", out.get("fabric_frac", 0.0))}
    bound = max(cand, key=cand.get)
    if out.get("wait_frac", 0.0) >= 0.6 and cand[bound] < 0.5:
        bound = "latency"
    out["bound"] = bound
    return out


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


def usable_cores():
    """CPUs this process may actually use: the cgroup quota (/sys/fs/cgroup/cpu.max = "quota period"; the GPU boxes show 256 CPUs
    and grant 16), else the affinity mask, else os.cpu_count()."""
    n = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = max(1, int(round(float(q) / float(per))))
    except Exception:
        pass
    try:
        a = len(os.sched_getaffinity(0))
        n = a if n is None else min(n, a)
    except Exception:
        pass
    return n or os.cpu_count() or 1


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
            "sample": "prefixes of the bench workload (%d bp) of %s reads (SURVEY 8(d): >= 1 M reads at -z all usable cores and -z 8; the -z 1 run is bounded to ~15 s); "
                      "%s -e 1 -t n -p f -b 0, wall time of the process incl. DB load; headline = first run" %
                      (read_len, " / ".join(str(r["reads"]) for r in runs), "rapsearch_Linux_2.15 -z T" if kind == "reference" else "oracle/rs_port (C restatement, 1 thread)"),
            "note": "'cores' = the -z of the headline run = the CPUs this process may use (cgroup quota /sys/fs/cgroup/cpu.max, else the affinity mask); host_cores = "
                    "os.cpu_count(), what the machine shows. RAPsearch2 2.15 stops scaling at about 8 threads (see runs), whatever the host has",
            "usable_cores": usable_cores(), "host_cores": os.cpu_count(), "runs": runs, "m8_md5_equals_gpu": all(r["m8_md5_equals_gpu"] for r in runs)}


def write_fastq(gen, n, L, path, gz, id0=0, id_width=0):
    """n reads of the bench workload as a FASTQ file (ids id0 .. id0 + n - 1, qualities 'I' with a '5' every tenth base: phred+33)."""
    import gzip
    import numpy as np
    reads = gen.single(n, L, first=(1 << 40) + id0).cpu().numpy()      # (indices far away from the resident set)
    w = max(id_width, len(str(id0 + n - 1)))
    rec = np.empty((n, 1 + w + 1 + L + 3 + L + 1), dtype=np.uint8)
    rec[:, 0] = ord("@")
    ids = np.arange(id0, id0 + n)
    for k in range(w):
        rec[:, w - k] = ord("0") + (ids // 10 ** k) % 10
    rec[:, 1 + w] = 10
    rec[:, 2 + w:2 + w + L] = reads
    rec[:, 2 + w + L:5 + w + L] = np.frombuffer(b"\n+\n", dtype=np.uint8)
    rec[:, 5 + w + L:5 + w + 2 * L] = ord("I")
    rec[:, 5 + w + L + 7:5 + w + 2 * L:10] = ord("5")          # (a character only phred+33 files hold: the offset detection stops at the first record)
    rec[:, -1] = 10
    if gz:
        with gzip.open(path, "wb", compresslevel=1) as f:
            f.write(rec.tobytes())
    else:
        rec.tofile(path)
    return os.path.getsize(path)


def e2e_rate(device, gen, n, L, gz):
    """File -> AGS through run_pipeline (native reader, HIP search, classification, estimate) on a FASTQ file of n reads of the
    bench workload written to the box's temp directory.  Returns reads/s of the whole call (file open to estimate)."""
    import contextlib
    import io
    from microbecensus_amd import microbe_census as mc
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "reads.fq" + (".gz" if gz else ""))
        size = write_fastq(gen, n, L, path, gz)
        walls = []
        for rep in range(2):                                       # the first call also allocates the engine's pools for this batch size
            args = {"seqfiles": [path], "device": device, "nreads": n, "read_length": L}
            t = time.time()
            with contextlib.redirect_stdout(io.StringIO()):
                res = mc.run_pipeline(args)
            walls.append(time.time() - t)
            if res is None:
                return None
        cold = None
        if not gz and n >= 2_000_000:
            # the reference's default use: ONE run_pipeline of 1 - 2 M reads per process (scripts/run_microbe_census.py:31) - the CLI in a
            # fresh process on the same file, wall time including the interpreter, HIP start-up, engine open and pool allocation
            cli = os.path.join(REPO, "scripts", "run_microbe_census.py")
            cw, ags = [], None
            for rep in range(3):
                outp = os.path.join(td, "cold%d.txt" % rep)
                t = time.time()
                rc = subprocess.call([sys.executable, cli, "-n", "2000000", path, outp], stdout=subprocess.DEVNULL)
                cw.append(round(time.time() - t, 3))
                if rc == 0 and os.path.exists(outp):
                    ags = [float(l.split("\t")[1]) for l in open(outp) if l.startswith("average_genome_size")][0]
            cold = {"what": "scripts/run_microbe_census.py -n 2000000 <plain FASTQ> in a fresh process, wall time of the process (three runs; the first may build the "
                            "per-user index cache, ~/.cache/microbecensus_amd)", "reads": 2_000_000, "wall_s_runs": cw, "wall_s": min(cw), "est_ags": ags}
    dt = walls[-1]
    out = {"reads": n, "file": "FASTQ" + (".gz" if gz else ""), "file_bytes": size, "wall_s": round(dt, 3), "reads_per_s": round(n / dt, 1),
           "first_call_wall_s": round(walls[0], 3), "sampled_reads": int(res[1]["sampled_reads"]), "est_ags": res[0], "sampler_seconds": res[1].get("_sampler_seconds")}
    if cold:
        out["cold_cli"] = cold
    return out


def write_fastq_c5(gen, n, L, path):
    """BASELINE configs[4]'s input at size - the recipe of tests/golden/c5_at_size.py, array by array instead of record by record:
    n reads of L bp of the 30 genomes (the two IUPAC letters the genomes hold become N: reverse_complement knows ACGTN only), phred+33
    qualities ~ N(34, 6) clipped to [20, 41], 5 % of the records with one base of quality 10, 2 % exact and 1 % reverse-complement
    copies of one of the first 5,000 records of their block of 150,000.  Returns (file bytes, records that are copies)."""
    import numpy as np
    BLOCK, POOL = 150_000, 5_000
    comp = np.arange(256, dtype=np.uint8)
    for a, b in zip(b"ACGTN", b"TGCAN"):
        comp[a] = b
    w = len(str(n - 1))
    copies = 0
    with open(path, "wb") as f:
        for b0 in range(0, n, BLOCK):
            m = min(BLOCK, n - b0)
            rng = np.random.default_rng(3000 + b0 // BLOCK)
            r = gen.single(m, L, first=(1 << 41) + b0).cpu().numpy().copy()
            r[(r == ord("Y")) | (r == ord("S"))] = ord("N")
            u = rng.random(m)
            idx = np.arange(m)
            src = (rng.random(m) * np.minimum(idx, POOL)).astype(np.int64)          # an EARLIER record of the block's first 5,000
            exact = (u < 0.02) & (idx > 0)
            rc = (u >= 0.02) & (u < 0.03) & (idx > 0)
            for i in np.nonzero(exact | rc)[0]:                                      # in order: a copy of a copy is a copy of the original (4,500 per block)
                r[i] = comp[r[src[i]][::-1]] if rc[i] else r[src[i]]
            copies += int(exact.sum() + rc.sum())
            q = np.clip(np.rint(rng.standard_normal((m, L), dtype=np.float32) * 6 + 34), 20, 41).astype(np.uint8) + 33
            low = np.nonzero(rng.random(m) < 0.05)[0]
            q[low, rng.integers(0, L, size=low.size)] = 33 + 10
            rec = np.empty((m, 1 + w + 1 + L + 3 + L + 1), dtype=np.uint8)
            rec[:, 0] = ord("@")
            ids = b0 + idx
            for k in range(w):
                rec[:, w - k] = ord("0") + (ids // 10 ** k) % 10
            rec[:, 1 + w] = 10
            rec[:, 2 + w:2 + w + L] = r
            rec[:, 2 + w + L:5 + w + L] = np.frombuffer(b"\n+\n", dtype=np.uint8)
            rec[:, 5 + w + L:5 + w + 2 * L] = q
            rec[:, -1] = 10
            f.write(rec.tobytes())
    return os.path.getsize(path), copies


def e2e_c5(device, gen, n, L=300):
    """BASELINE configs[4] end to end: run_pipeline(file -> AGS) on a plain FASTQ of n records of 300 bp with min_quality 20 and
    filter_dups (the reference's -q 20 -d; process_seqfile microbe_census.py:328-367 - duplicates are tested before the quality filter,
    only accepted reads enter the set).  Wall time of the second of two calls, the sampler's own phase timers beside it."""
    import contextlib
    import io
    from microbecensus_amd import microbe_census as mc
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "c5.fq")
        size, copies = write_fastq_c5(gen, n, L, path)
        walls, res = [], None
        for rep in range(2):
            args = {"seqfiles": [path], "device": device, "nreads": n, "read_length": L, "min_quality": 20, "filter_dups": True}
            t = time.time()
            with contextlib.redirect_stdout(io.StringIO()):
                res = mc.run_pipeline(args)
            walls.append(time.time() - t)
            if res is None:
                return None
        # the same file without -d and without -q: what the duplicate filter costs the call
        args = {"seqfiles": [path], "device": device, "nreads": n, "read_length": L}
        t = time.time()
        with contextlib.redirect_stdout(io.StringIO()):
            plain = mc.run_pipeline(args)
        plain_wall = time.time() - t
    dt = walls[-1]
    return {"what": "BASELINE configs[4]: run_pipeline(file -> AGS) on a plain FASTQ of %d records of %d bp (recipe of tests/golden/c5_at_size.py: 5 %% with a base of quality 10, "
                    "2 %% exact + 1 %% reverse-complement copies) with min_quality=20, filter_dups=True; wall time of the second of two calls" % (n, L),
            "records": n, "read_len": L, "file_bytes": size, "copies_written": copies, "wall_s": round(dt, 3), "reads_per_s": round(n / dt, 1), "first_call_wall_s": round(walls[0], 3),
            "sampled_reads": int(res[1]["sampled_reads"]), "est_ags": res[0], "sampler_seconds": res[1].get("_sampler_seconds"),
            "same_file_without_q_and_d": None if plain is None else {"wall_s": round(plain_wall, 3), "reads_per_s": round(n / plain_wall, 1), "sampled_reads": int(plain[1]["sampled_reads"]),
                                                                      "sampler_seconds": plain[1].get("_sampler_seconds")}}


def e2e_distributed(gen, n, L, rank, world, local, rdev):
    """File -> AGS over all ranks (microbecensus_amd.distributed.run_pipeline_distributed) on a plain FASTQ: every rank samples and
    searches its own slices of the file; one all_reduce of the per-family sums.  Wall time from before the call to after it on every
    rank, the maximum over the ranks."""
    import contextlib
    import io
    import shutil
    import torch
    import torch.distributed as dist
    from microbecensus_amd import distributed as mcd
    box = [None]
    if rank == 0:
        td = tempfile.mkdtemp(prefix="mc_e2e_")
        box[0] = os.path.join(td, "reads.fq")
        write_fastq(gen, n, L, box[0], False)
    dist.broadcast_object_list(box, src=0)
    walls, est = [], None
    for rep in range(2):
        dist.barrier()
        t = time.time()
        with contextlib.redirect_stdout(io.StringIO()):
            est, a = mcd.run_pipeline_distributed({"seqfiles": [box[0]], "nreads": n, "read_length": L}, device=local)
        dist.barrier()
        w = torch.tensor([time.time() - t], dtype=torch.float64, device=rdev)
        dist.all_reduce(w, op=dist.ReduceOp.MAX)
        walls.append(float(w.item()))
    out = {"what": "run_pipeline_distributed(file -> AGS) over %d ranks on a plain FASTQ: every rank samples its own slices of the file (mc_reader_open_range) and searches "
                   "them, the head-take and the read indices come from the exchanged counts, one all_reduce of the per-family sums; wall time of the second of two "
                   "calls, maximum over the ranks" % world, "reads": n, "file": "FASTQ", "wall_s": round(walls[-1], 3), "reads_per_s": round(n / walls[-1], 1),
           "first_call_wall_s": round(walls[0], 3), "sampled_reads": int(a["sampled_reads"]), "est_ags": est}
    # the same on a FASTQ.gz: every rank inflates its own chunk slices, the 32 KB windows handed along the ranks.  The file is
    # written by all ranks at once - a gzip member each, concatenated by rank 0 - because one Python gzip writer takes 4 s per 1 M reads
    ngz = min(n, 2_000_000 * world)
    share = ngz // world
    ngz = share * world
    td = os.path.dirname(box[0])
    part = os.path.join(td, "part%d.gz" % rank)
    write_fastq(gen, share, L, part, True, id0=rank * share, id_width=len(str(ngz - 1)))
    dist.barrier()
    gzp = os.path.join(td, "reads.fq.gz")
    if rank == 0:
        with open(gzp, "wb") as f:
            for r in range(world):
                with open(os.path.join(td, "part%d.gz" % r), "rb") as g:
                    shutil.copyfileobj(g, f, 1 << 24)
    walls, est = [], None
    try:
        for rep in range(2):
            dist.barrier()
            t = time.time()
            with contextlib.redirect_stdout(io.StringIO()):
                est, a = mcd.run_pipeline_distributed({"seqfiles": [gzp], "nreads": ngz, "read_length": L}, device=local)
            dist.barrier()
            w = torch.tensor([time.time() - t], dtype=torch.float64, device=rdev)
            dist.all_reduce(w, op=dist.ReduceOp.MAX)
            walls.append(float(w.item()))
    except BaseException as e:                                     # noqa: BLE001 - (run_pipeline_distributed raises on every rank or on none: the line survives a failing leg)
        out["gz"] = {"error": "%s: %s" % (type(e).__name__, e)}
        dist.barrier()
        if rank == 0:
            shutil.rmtree(td, ignore_errors=True)
        return out
    out["gz"] = {"what": "the same on a FASTQ.gz of %d gzip members: every rank inflates its own slices of 32 chunks of 1 MB, the windows handed along the ranks, "
                         "and samples the records that start in its text (nothing dealt by rank 0: %s)" % (world, mcd.run_pipeline_distributed.last_trace is None),
                 "reads": ngz, "file": "FASTQ.gz", "file_bytes": os.path.getsize(gzp) if rank == 0 else None, "wall_s": round(walls[-1], 3), "reads_per_s": round(ngz / walls[-1], 1),
                 "first_call_wall_s": round(walls[0], 3), "sampled_reads": int(a["sampled_reads"]), "est_ags": est}
    dist.barrier()
    if rank == 0:
        shutil.rmtree(td, ignore_errors=True)
    return out


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
    ap.add_argument("--cpu-sample", type=int, default=1_000_000, help="reads of the cpu_baseline runs at -z <usable cores> and -z 8 (SURVEY 8(d): >= 1 M; ~70 s each); the -z 1 run takes a 75th of it")
    ap.add_argument("--cpu-full", action="store_true", help="cpu_baseline on the full sample at -z 1 too (~6 min more)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ags-check", action="store_true", help="skip the run_pipeline AGS comparison on the reference's own inputs")
    ap.add_argument("--e2e-reads", type=int, default=20_000_000, help="reads of the end-to-end (file -> AGS) measurement (plain FASTQ; a fifth of it, at least 4 M, for .gz); 0 = skip")
    ap.add_argument("--c5-reads", type=int, default=2_000_000, help="records of the BASELINE configs[4] leg (e2e.c5: 300 bp FASTQ, -q 20 -d, file -> AGS); 0 = skip")
    ap.add_argument("--no-best-only-leg", action="store_true", help="skip the extra timed leg with mc_set_best_hits_only (what run_pipeline runs)")
    ap.add_argument("--no-reference-pattern", action="store_true", help="skip the untimed launch of the counting form of the seed kernel (roofline.reference_pattern); the profiling tools do")
    ap.add_argument("--one-at-a-time", action="store_true", help="mc_run_range per step (the device idles while the host sums the best hits up) instead of mc_range_end / mc_range_begin / results")
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

    def collect_results(collect=True):
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

    def run_steps(k, collect=True):
        """k steps.  A step is one batch through the whole path - search, best hits, per-family accumulators (and their all_reduce).
        As mc_search / mc_search_files do with their batches, the front of step i + 1 (translation, seeds) is enqueued before the host
        looks at the results of step i (mc_range_end, mc_range_begin, results): the device does not idle while the host sums up.
        --one-at-a-time: mc_run_range per step, as rounds 1 - 3 measured."""
        acc = {}
        def add(st):
            for key, v in st.items():
                acc[key] = acc.get(key, 0) + v
        if args.one_at_a_time:
            for i in range(k):
                b = i % nres
                eng.run_range(b * args.batch, args.batch, first_read_id=b * args.batch)
                add(collect_results(collect))
            return acc
        for i in range(k + 1):
            if i > 0:
                eng.range_end()
            if i < k:
                b = i % nres
                eng.range_begin(b * args.batch, args.batch, first_read_id=b * args.batch)
            if i > 0:
                add(collect_results(collect))
        return acc

    run_steps(args.warmup, collect=False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.time()
    acc = run_steps(K)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.time() - t0
    # the same steps with mc_set_best_hits_only (run_pipeline's path: only the reads that can be classified are ranked, no rows)
    dt_only, st_only = None, None
    if not args.no_best_only_leg:
        eng.set_best_hits_only(True)
        run_steps(1, collect=False)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t1 = time.time()
        st_only = run_steps(K, collect=False)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt_only = time.time() - t1
        eng.set_best_hits_only(False)
        if world > 1:
            t2 = torch.tensor([dt_only], dtype=torch.float64, device=rdev)
            dist.all_reduce(t2, op=dist.ReduceOp.MAX)
            dt_only = float(t2.item())
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
    # A yardstick that does not move with the implementation: what the REFERENCE's seed stage would read for one batch -
    # CHashSearch::Searching@0x415050 looks every seed's bucket up (8 B: start and size), binary-searches the bucket's suffix keys (2 B per
    # key read: lower_bound / upper_bound, ExtendSeq2Set 0x413bd2-0x414aa1), and reads posting, subject offset and the residues around the
    # seed of every seed hit (4 + 4 + 12 B); the frames once (6 x L/3 B).  ONE untimed launch of the counting form of the seed kernel
    # (mc_set_counting: it searches every probe the reference searches instead of asking its filters, and counts) on batch 0.
    ref_pattern = None
    if rank == 0 and not args.no_reference_pattern:
        eng.set_counting(True)
        eng.run_range(0, args.batch, first_read_id=0)
        cst = eng.stats()
        eng.set_counting(False)
        if cst["bucket_lookups"] > 0:
            rp_bytes = 8 * cst["bucket_lookups"] + 2 * cst["key_probes"] + 20 * cst["seed_tasks"] + 6 * (L // 3) * args.batch
            ref_pattern = {"bytes_per_read": round(rp_bytes / args.batch, 1), "bucket_lookups_per_read": round(cst["bucket_lookups"] / args.batch, 2),
                           "key_reads_per_read": round(cst["key_probes"] / args.batch, 2), "seed_hits_per_read": round(cst["seed_tasks"] / args.batch, 2),
                           "counting_launch_ms": round(cst["ms_seed"], 3), "_bytes_per_launch": rp_bytes,
                           "formula": "8 B x bucket lookups + 2 B x key reads of the binary searches + 20 B x seed hits + 6 x (L / 3) B of frames, per read; counted by one untimed "
                                      "launch of k_enumerate_count on batch 0"}
    rccl = None
    if world > 1:
        # evidence that the collective really spanned `world` ranks: every rank contributes rank + 1 to a sum (must be N (N + 1) / 2) and its
        # LOCAL_RANK's device index to a bit mask; the crc of the reduced per-family vector the estimate is made from
        import zlib
        probe = torch.tensor([rank + 1, 1 << local], dtype=torch.int64, device=rdev)
        dist.all_reduce(probe)
        ver = None
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version()) if args.backend == "nccl" else None
        except Exception:
            pass
        rccl = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "nccl_version": ver, "rank_sum": int(probe[0].item()), "rank_sum_expected": world * (world + 1) // 2,
                "device_mask": int(probe[1].item()), "reduced_vector_crc32": zlib.crc32(np.concatenate([tot_hits, tot_aln, tot_bylen.ravel()]).tobytes())}
    e2e_multi = None
    if world > 1 and args.e2e_reads > 0 and gen is not None:
        eng.attach(0, 0)
        e2e_multi = e2e_distributed(gen, min(args.e2e_reads, 4_000_000 * world), L, rank, world, local, rdev)

    if rank == 0:
        reads_total = args.batch * K * world
        # dominant kernel = the one with the largest accumulated HIP-event time (rank 0's events, on the library's own stream)
        kern = {"k_translate_seg": acc["ms_translate"], "k_enumerate": acc["ms_seed"], "k_eval_seeds": acc["ms_eval"], "k_gapped": acc["ms_gapped"],
                "sort": acc["ms_sort"], "k_finish": acc["ms_finish"]}
        SEQ = {"k_translate_seg": "ms_translate", "k_enumerate": "ms_seed", "k_eval_seeds": "ms_eval", "k_gapped": "ms_gapped", "sort": "ms_sort", "k_finish": "ms_finish"}
        kseq = {k: acc[m] / K for k, m in SEQ.items()}              # ms per step: HIP events around each stage on the library's stream (one kernel at a time)
        dom = max(kseq, key=kseq.get)
        n_batch = args.batch
        hits, hsps, gtasks, rows = acc["seed_tasks"] / K, acc["hsps"] / K, acc["gap_tasks"] / K, acc["rows"] / K
        # What the seed kernel asks of the index per launch (mc_stats.seed_*, counted by the timed kernel; DESIGN.md 5 "index touches"):
        # 9-mer filter words (4 B), wildcard filter lines (32 B), pair filter blocks (16 B), bucket record + key group per probe that
        # survives the filters (32 + 16 B).  (Round 5: a seed hit's posting and its subject's offsets are no longer read here - the hit
        # record holds the posting's index, k_eval_seeds fetches posting, place in the residue array and rest of the subject in one
        # 8-byte load: those 8 B per hit moved to residue_touch.)
        asks = {k: acc[k] / K for k in ("seed_exact_asks", "seed_wild_asks", "seed_pair_asks", "seed_probes")}
        index_touch = 4 * asks["seed_exact_asks"] + 32 * asks["seed_wild_asks"] + 16 * asks["seed_pair_asks"] + 48 * asks["seed_probes"]
        # ... and the evaluation kernel of the residues
        residue_touch = (24 + 24 + 8) * hits                              # 24 bytes around the seed in the frame and in the subject, the 8-byte posting entry (MC_POST8)
        per_launch = {
            # ALGORITHMIC bytes per launch = the arrays a kernel must read and write, each once, plus - for the two kernels that gather
            # from the index - the items they ask of it, each at its own size.
            "k_translate_seg": n_batch * (L + 6 * (L // 3)),                 # bases in, six frames out
            "k_enumerate": n_batch * 6 * (L // 3) + 16 * hits + index_touch,  # frames in, seed hits out (16 B), index touches
            "k_eval_seeds": 16 * hits + residue_touch + 48 * hsps + 28 * gtasks,   # seed hits in, residues around them, HSPs (48 B) and gap tasks (28 B) out
            "k_gapped": (28 + 32 + 48) * gtasks,                             # gap tasks in, two flank results (16 B) and an HSP out per task
            "sort": hsps * (16 + 20 + 20 + 96 * 0.4),                        # stage C: keys and place words in, 20 B binned and read again, the records of the marked reads' HSPs (~40 %) fetched and written
            "k_finish": hsps * 48 + rows * 64,                               # HSPs in, m8 rows out
        }
        if ref_pattern is not None and kseq["k_enumerate"] > 0:
            rp_rate = ref_pattern.pop("_bytes_per_launch") / (kseq["k_enumerate"] * 1e-3) / 1e9
            ref_pattern.update({"timed_seed_kernel_ms": round(kseq["k_enumerate"], 3), "disposed_GBps": round(rp_rate, 1), "frac_of_hbm_peak": round(rp_rate / HBM_PEAK_GBS, 4),
                                "own_asks_bytes_per_read": round(per_launch["k_enumerate"] / n_batch, 1),
                                "note": "disposed_GBps = the reference pattern's bytes of one batch / the TIMED seed kernel's duration: above the HBM peak means the kernel answers "
                                        "most of what the reference would read from its filters instead (own_asks_bytes_per_read is what it really asks)"})
        prof = load_profile(L)
        der = {k: derived(prof, "k_enumerate_t0" if k == "k_enumerate" else k) for k in kseq}
        d_dom = der.get(dom)
        ach = per_launch[dom] / (kseq[dom] * 1e-3) / 1e9
        # the whole device pipeline of one batch as one "launch": SURVEY 8(d)'s own definition, achieved = reads/s x A(L) - a LEGACY
        # figure: its A(L) assumes whole-bucket visits the engine does not perform
        pipe_gbs = (SURVEY_A[L] * (reads_total / world) / dt / 1e9) if L in SURVEY_A else None
        agg = mcd.aggregate_from_accumulators(tot_hits, tot_aln, tot_bylen, fams, mc.find_opt_pars(None, L))
        try:
            est = mc.estimate_average_genome_size({"read_length": L, "sampled_reads": reads_total, "verbose": False}, None, agg)
        except BaseException:
            est = None
        # HBM bytes of the dominant kernel per launch of THIS run's batch: the profile's counters are per launch of its own batch size;
        # the pipeline is linear in the number of reads, so only the reads-per-launch ratio is applied - never another read length
        traffic_dom = None
        if d_dom and d_dom.get("fabric_bytes_per_launch") is not None and d_dom.get("reads_per_launch"):
            traffic_dom = d_dom["fabric_bytes_per_launch"] * n_batch / d_dom["reads_per_launch"]
        # north_star's second target: >= 40 % of the HBM roofline on the extension kernel(s) - stated, and missed: both are bound by
        # VALU issue (integer DP / X-drop loops with 28 - 37 of 64 lanes active), not by bytes
        ext = {}
        for k in ("k_eval_seeds", "k_gapped"):
            if kseq.get(k, 0) > 0:
                a = per_launch[k] / (kseq[k] * 1e-3) / 1e9
                ext[k] = {"algorithmic_GBps": round(a, 1), "frac_of_hbm_peak": round(a / HBM_PEAK_GBS, 4), "counter_fabric_frac": (der.get(k) or {}).get("fabric_frac"),
                          "counter_fabric_frac_lower": (der.get(k) or {}).get("fabric_frac_lower"), "bound": (der.get(k) or {}).get("bound"),
                          "issue_roofline": (der.get(k) or {}).get("issue_roofline")}
        ext_best = max([v["frac_of_hbm_peak"] for v in ext.values()] or [0.0])
        ev_req = None
        try:                                                     # (L1 -> L2 read requests per seed hit of the committed profile of this read length: what the model prices)
            evc = stage_counters(prof, "k_eval_seeds")
            if evc and evc.get("TCP_TCC_READ_REQ_sum") and (der.get("k_eval_seeds") or {}).get("reads_per_launch"):
                ev_req = round(evc["TCP_TCC_READ_REQ_sum"] / (hits * der["k_eval_seeds"]["reads_per_launch"] / n_batch), 2)
        except Exception:                                         # noqa: BLE001
            ev_req = None
        elc = eval_line_model(gather_ceiling(), n_batch, hits, hsps + gtasks, kseq["k_eval_seeds"], ev_req)
        if elc is not None and "k_eval_seeds" in ext:              # (as for the seed kernel: the label follows this run's measurement)
            ext["k_eval_seeds"]["bound_by_counter_fractions"] = ext["k_eval_seeds"]["bound"]
            if elc["frac"] >= 0.8:
                ext["k_eval_seeds"]["bound"] = "fabric"
        # What binds the dominant kernel, from THIS run's measurements (no label or prose that cannot change): the seed kernel is
        # called bound by the fabric's scattered lines when its own ask counters, priced at the machine's calibrated scattered-line rates
        # (profiles/rNN_gather_ceiling.json), account for at least 0.8 of its live duration; otherwise the largest counter fraction of the
        # committed profile of this read length decides (null without one).
        slc = seed_line_model(gather_ceiling(), L, n_batch, asks, hits, kseq["k_enumerate"])
        bound_label, bound_evidence = (d_dom or {}).get("bound"), None
        if dom == "k_enumerate" and slc is not None:
            if slc["frac"] >= 0.8:
                bound_label = "fabric"
            bound_evidence = ("this run: %.1f scattered lines asked per read, priced at the calibrated rates of %s = %.3f ms per launch against %.3f ms measured (frac %.3f; >= 0.8 reads as "
                              "'the kernel takes the time the memory system needs to deliver its lines')" % (slc["lines_asked_per_read"], slc["calibration"], slc["model_ms_per_launch"], slc["measured_ms_per_launch"], slc["frac"]))
        out = {
            "metric": METRIC, "value": round(reads_total / dt, 1), "unit": "reads/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": round(dt / K * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/int32",
            "data": "synthetic",
            "config": {"workload": "%s; %d reads/step/GPU, %d steps, %d distinct reads resident in HBM per GPU" % (wl, n_batch, K, per_rank),
                       "read_len": L, "batch": n_batch, "parallelism": ("1 GPU, no collective" if world == 1 else "reads sharded over %d GPUs, %s all_reduce of per-family accumulators per step" % (world, "RCCL" if args.backend == "nccl" else "gloo (plumbing test: all ranks on one GPU)")),
                       "marker_db": "%d proteins / %d families" % (len(names), len(fams)),
                       "classified_reads": int(tot_hits.sum()), "classified_per_read": round(float(tot_hits.sum()) / reads_total, 6),
                       "rows_per_read": round(job["rows"] / reads_total, 4), "reads_with_rows": round(job["reads_with_rows"] / reads_total, 5),
                       "hsps_per_read": round(job["hsps"] / reads_total, 3), "gapped_extensions_per_read": round(job["gap_tasks"] / reads_total, 3),
                       "seed_hits_per_read": round(job["seed_tasks"] / reads_total, 2), "ags_estimate_of_workload": est,
                       "steps_issued": "mc_run_range per step" if args.one_at_a_time else "mc_range_end(i), mc_range_begin(i + 1), results of i: the front of the next step is enqueued before the host sums up (as mc_search / mc_search_files do with their batches)",
                       # HIP events on the library's own streams around each stage, timed region (one kernel at a time)
                       "kernel_ms_per_step": {k: round(v / K, 3) for k, v in kern.items()},
                       "sum_kernel_ms_per_step": round(sum(kern.values()) / K, 3)},
            "roofline": {"kernel": dom, "bound": bound_label, "bound_by_counter_fractions": (d_dom or {}).get("bound"), "nominal_bound": "hbm",
                         "bound_evidence": bound_evidence,
                         "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
                         "traffic": (None if traffic_dom is None else round(traffic_dom, 0)),
                         # flat copies of the nested yardsticks below (a reader that keeps only the scalars of this object keeps these: VERDICT r05 item 7)
                         "reference_pattern_bytes_per_read": (ref_pattern or {}).get("bytes_per_read"), "reference_pattern_frac_of_hbm_peak": (ref_pattern or {}).get("frac_of_hbm_peak"),
                         "legacy_survey_A_bytes_per_read": SURVEY_A.get(L), "legacy_survey_A_pipeline_GBps": (None if pipe_gbs is None else round(pipe_gbs, 2)),
                         "scattered_line_ceiling_frac": (slc or {}).get("frac"), "extension_kernel_hbm_frac_best": round(ext_best, 4), "extension_kernel_scattered_line_ceiling_frac": (elc or {}).get("frac"),
                         "fabric_frac": (d_dom or {}).get("fabric_frac"), "fabric_frac_lower": (d_dom or {}).get("fabric_frac_lower"), "issue_frac": (d_dom or {}).get("issue_frac"),
                         "issue_roofline": (d_dom or {}).get("issue_roofline"), "reference_pattern": ref_pattern,
                         "scattered_line_ceiling": slc, "extension_kernel_scattered_line_ceiling": elc,
                         "salu_frac": (d_dom or {}).get("salu_frac"), "wait_frac": (d_dom or {}).get("wait_frac"), "wave_residency": (d_dom or {}).get("wave_residency"),
                         "valu_lanes_of_64": (d_dom or {}).get("valu_lanes_of_64"), "l2_hit_rate": (d_dom or {}).get("l2_hit_rate"),
                         "kernel_ms_per_step": round(kseq[dom], 3), "algorithmic_bytes_per_read": round(per_launch[dom] / n_batch, 1),
                         "index_touch_bytes_per_read": round(index_touch / n_batch, 1),
                         "index_touch_items_per_read": {"filter_words_4B": round(asks["seed_exact_asks"] / n_batch, 2), "wildcard_lines_32B": round(asks["seed_wild_asks"] / n_batch, 2),
                                                        "pair_blocks_16B": round(asks["seed_pair_asks"] / n_batch, 2), "records_and_key_groups_48B": round(asks["seed_probes"] / n_batch, 2),
                                                        "postings_and_offsets_8B": 0.0},
                         "fabric_amplification": (None if traffic_dom is None else round(traffic_dom / per_launch[dom], 2)),
                         "fabric_amplification_lower": (None if not (d_dom and d_dom.get("fabric_frac_lower") and d_dom.get("fabric_frac")) else round(traffic_dom / per_launch[dom] * d_dom["fabric_frac_lower"] / d_dom["fabric_frac"], 2)),
                         "extension_kernel_hbm_frac": {"target": 0.40, "met": bool(ext_best >= 0.40), "best": round(ext_best, 4), "kernels": ext,
                                                       "note": "north_star asks for >= 40 % of the HBM roofline on the extension kernel; the ungapped (k_eval_seeds) and gapped (k_gapped stage) "
                                                               "extensions are not bound by bytes: k_eval_seeds by the scattered lines of its hits (extension_kernel_scattered_line_ceiling), the gapped stage by VALU issue: the target is missed"},
                         "profile": (None if prof is None else prof["files"]), "fetch_calibration": (None if prof is None else prof.get("calibration")),
                         "basis": "kernel = the stage with the largest HIP-event time per step. achieved = ALGORITHMIC bytes of one launch (the arrays the kernel "
                                  "must read and write, each once, plus the items the seed kernel asks of the index at their own sizes - index_touch_bytes_per_read, "
                                  "from mc_stats.seed_*: DESIGN.md 5) / the live HIP-event duration of that launch; frac = achieved / 8 TB/s; fabric_amplification = "
                                  "traffic / those bytes (one 64- or 128-byte line moves per 4 .. 48-byte item). Everything else comes from the committed rocprofv3 profile of THIS read length (profile; null "
                                  "when none is committed - a profile of another length is never scaled): traffic = (fetch_factor x FETCH_SIZE + WRITE_SIZE) KiB x 1024 "
                                  "per profiled launch x (this batch / profiled batch), fetch_factor 2 for streaming reads (MI355X_MICROARCH.md) and the gather "
                                  "calibration's (fetch_calibration) for the seed kernels; fabric_frac = that / profiled duration / 8 TB/s (L2-miss traffic, mostly Infinity-Cache hits: "
                                  "the index is 110 MB - rounds 1 - 4 called it hbm_frac; fabric_frac_lower: the same with one 64-byte half line per scattered request); "
                                  "reference_pattern = what CHashSearch::Searching / ExtendSeq2Set would read for this batch, counted by ONE untimed launch of the counting form of "
                                  "the seed kernel (mc_set_counting) - a yardstick that does not move with this implementation's filters; issue_roofline = issue_frac x valu_lanes_of_64 / 64; issue_frac = SQ_INSTS_VALU x 4 / "
                                  "(1024 SIMDs x kernel cycles), salu_frac = SQ_INSTS_SALU / (256 scalar units x kernel cycles) - the two issue side by side and are "
                                  "not added -, kernel cycles = profiled duration x 2.4 GHz; wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES; valu_lanes_of_64 = "
                                  "SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU; bound = the largest of issue_frac, salu_frac and fabric_frac (fabric_frac_lower where there is one; 'latency' when all are "
                                  "below 0.5 and wait_frac >= 0.6). No kernel of this path is HBM bound: the nominal roofline (nominal_bound) is kept because "
                                  "the metric asks for it.",
                         "all_kernels": {k: dict({"ms_per_step": round(kseq[k], 3), "algorithmic_GBps": round(per_launch[k] / (kseq[k] * 1e-3) / 1e9, 2)},
                                                 **({} if not der.get(k) else {m: der[k][m] for m in ("bound", "issue_frac", "issue_roofline", "salu_frac", "wait_frac", "wave_residency", "fabric_frac", "fabric_frac_lower", "valu_lanes_of_64", "lds_bank_conflict_frac", "l2_hit_rate", "profiled_ms") if m in der[k]}))
                                         for k in kseq if kseq[k] > 0},
                         "legacy_survey_A": {"bytes_per_read": SURVEY_A.get(L), "pipeline_GBps": (None if pipe_gbs is None else round(pipe_gbs, 2)),
                                             "note": "SURVEY.md 8(d) priced the whole path at A(L) assuming whole-bucket visits the engine does not perform (DESIGN.md 2): kept as a labelled legacy figure, not a fraction of anything"}},
        }
        if dt_only is not None:
            out["classification_only"] = {"what": "the same steps with mc_set_best_hits_only (what run_pipeline runs when not verbose): every HSP is made, only the reads "
                                                  "that have an HSP passing their family's thresholds are sorted and finished, no m8 rows; identical best hits",
                                          "value": round(reads_total / dt_only, 1), "unit": "reads/s", "ms_per_step": round(dt_only / K * 1e3, 3),
                                          "kernel_ms_per_step": {k: round(st_only[m] / K, 3) for k, m in SEQ.items()}}
        if world == 1 and not args.no_cpu_baseline:
            cores = usable_cores()
            plan = [(cores, args.cpu_sample), (8, args.cpu_sample), (1, args.cpu_sample if args.cpu_full else max(1000, args.cpu_sample // 75))]
            if cores == 8:
                plan.pop(1)
            ns = min(max(n for _, n in plan), per_rank)
            out["cpu_baseline"] = cpu_baseline(eng, reads[:ns].cpu().numpy(), L, [(t, min(n, ns)) for t, n in plan])
        if world == 1 and not args.no_ags_check:
            out["config"]["ags_abs_error_vs_reference"] = ags_abs_error(local)
        if world == 1 and args.e2e_reads > 0 and gen is not None:
            out["e2e"] = {"what": "run_pipeline(file -> AGS): native reader beside the HIP search (mc_search_files), classification, estimate; wall time of the second "
                                  "of two calls on the same file (first_call_wall_s includes the one-time pool allocation); .gz is inflated by several workers (csrc/mc_pgzip.h) as far as the CPUs the process may use allow (cgroup quota)",
                          "plain": e2e_rate(local, gen, args.e2e_reads, L, gz=False), "gz": e2e_rate(local, gen, max(1, min(args.e2e_reads, max(args.e2e_reads // 5, 4_000_000))), L, gz=True)}
        if world == 1 and args.c5_reads > 0 and gen is not None:
            out.setdefault("e2e", {})["c5"] = e2e_c5(local, gen, args.c5_reads)
        if e2e_multi is not None:
            out["e2e"] = e2e_multi
        if rccl is not None:
            out["rccl"] = rccl
        print(json.dumps(out))
        sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
