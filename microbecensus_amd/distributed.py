"""Multi-GPU layer: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on MI355X).

The path shards embarrassingly: the accepted-read stream (after the sequential sampling / QC / duplicate
semantics of process_seqfile, which stay on the host of rank 0) is cut into contiguous blocks, one per rank
(one scatter), the
29 MB marker index is replicated in every GPU's HBM, and nothing is exchanged while searching.  The only
exchange step is the final reduction of the per-family accumulators that aggregate_hits() needs
(reference microbe_census.py:462-472):

    hits[f]            number of classified reads                       int64
    aln_sum[f]         sum of alignment lengths                         int64
    aln_by_len[f, t]   sum of alignment lengths per target length t     int64   (for aln_stat == 'cov')

All integers, so the all_reduce is exact and order independent; cov sums are finished on the host as
sum_t aln_by_len[f, t] / t, which equals the reference's sum of aln/target_len up to double rounding
(<= 1e-12 relative, the tolerance BASELINE.md states for 'cov' families).
"""
import numpy as np

MAX_TARGET_LEN = 2048


def shard_bounds(n_items, rank, world):
    """Contiguous block [lo, hi) of rank `rank`; blocks differ by at most one item and keep read ids global."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def family_accumulators(best_hits, nfam):
    """best_hits: structured array with fields family, aln, target_len -> (hits, aln_sum, aln_by_len)."""
    hits = np.bincount(best_hits["family"], minlength=nfam).astype(np.int64)
    aln_sum = np.bincount(best_hits["family"], weights=best_hits["aln"], minlength=nfam).astype(np.int64)
    aln_by_len = np.zeros((nfam, MAX_TARGET_LEN), dtype=np.int64)
    np.add.at(aln_by_len, (best_hits["family"], best_hits["target_len"]), best_hits["aln"])
    return hits, aln_sum, aln_by_len


def all_reduce_accumulators(hits, aln_sum, aln_by_len, device=None):
    """Sum the three accumulators over all ranks (RCCL when the process group is nccl, gloo on CPU)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return hits, aln_sum, aln_by_len
    flat = np.concatenate([hits.ravel(), aln_sum.ravel(), aln_by_len.ravel()])
    t = torch.from_numpy(flat)
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    flat = t.cpu().numpy()
    n = hits.size
    return flat[:n].reshape(hits.shape), flat[n:2 * n].reshape(aln_sum.shape), flat[2 * n:].reshape(aln_by_len.shape)


def aggregate_from_accumulators(hits, aln_sum, aln_by_len, families, optpars):
    """agg_hits {family: float} as aggregate_hits() would return it (families without hits are absent)."""
    agg = {}
    lens = np.arange(MAX_TARGET_LEN, dtype=np.float64)
    lens[0] = 1.0
    for i, fam in enumerate(families):
        if hits[i] == 0:
            continue
        stat = optpars[fam]["aln_stat"]
        if stat == "hits":
            agg[fam] = float(hits[i])
        elif stat == "aln":
            agg[fam] = float(aln_sum[i])
        else:
            agg[fam] = float(np.sum(aln_by_len[i] / lens))
    return agg


def scatter_reads(reads, n_total, read_len, rank, world, device=None):
    """Rank 0 holds the accepted reads (n_total x read_len uint8); every rank receives its contiguous block
    [rank * per, min((rank + 1) * per, n_total)), per = ceil(n_total / world), with ONE scatter (RCCL over xGMI when the group
    is nccl: the blocks travel GPU to GPU; gloo: host memory).  Returns (block, lo): block is a torch uint8 tensor of
    (hi - lo) * read_len bytes on `device` (nccl) or a numpy array (gloo)."""
    import torch
    import torch.distributed as dist
    per = -(-n_total // world) if n_total else 0
    lo, hi = min(rank * per, n_total), min((rank + 1) * per, n_total)
    if world == 1:
        return reads, 0
    on_gpu = dist.get_backend() == "nccl"
    dev = device if on_gpu else torch.device("cpu")
    recv = torch.empty(per * read_len, dtype=torch.uint8, device=dev)
    chunks = None
    if rank == 0:
        flat = torch.from_numpy(np.ascontiguousarray(reads).reshape(-1))
        if on_gpu:
            flat = flat.to(dev)
        pad = per * world * read_len - flat.numel()
        if pad:
            flat = torch.cat([flat, torch.zeros(pad, dtype=torch.uint8, device=flat.device)])
        chunks = list(flat.split(per * read_len))
    if per:
        dist.scatter(recv, chunks, src=0)
    block = recv[: (hi - lo) * read_len]
    return (block if on_gpu else block.numpy().reshape(hi - lo, read_len)), lo


def run_pipeline_distributed(args, device=None):
    """run_pipeline() over all ranks of the initialised torch.distributed group (one process per GPU; backend "nccl" = RCCL
    on MI355X, or gloo).  Rank 0 runs the (sequential, deterministic) sampler once and scatters contiguous blocks of the
    accepted reads; every rank searches its block on its GPU with global read ids, the per-family integer accumulators are
    summed with ONE all_reduce, and every rank finishes the estimate from the same sums.  Returns (est_ags, args) like
    run_pipeline; hits are integers and the 'cov' sums are finished from exact integer sums, so the result does not depend on
    the number of ranks (<= 1e-12 relative against the single-process sum order)."""
    import os
    import sys
    import torch
    import torch.distributed as dist
    from . import _native
    from . import microbe_census as mc
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
    paths = mc.get_relative_paths(args)                                        # (mkstemp: every rank has its own temp file)
    mc.check_paths(paths)
    mc.check_input(args)
    mc.impute_missing_args(args)
    mc.check_arguments(args)
    args["verbose"] = bool(args.get("verbose")) and rank == 0
    nccl = dist.is_initialized() and dist.get_backend() == "nccl"
    dev = torch.device("cuda", device) if nccl else None
    try:
        L = args["read_length"]
        reads, head = None, [0, 0, ""]
        if rank == 0:                                                          # the sampler: once, on rank 0
            try:
                reads, st = _native.sample_reads(args["seqfiles"], L, args["nreads"], args["file_type"] == "fastq", args.get("quality_offset") or 0,
                                                 args["min_quality"], args["mean_quality"], args["max_unknown"], args["filter_dups"])
                head = [int(st["sampled"]), int(st["bases"]) if st.get("exhausted") else -1, ""]
            except _native.ReferenceError_ as e:
                head = [-1, -1, str(e)]
        if world > 1:
            dist.broadcast_object_list(head, src=0)
        if head[0] < 0:
            raise Exception(head[2])                                           # the reference raises inside run_pipeline
        n_total = head[0]
        if n_total == 0:
            sys.exit("\nError! No reads remaining after filtering!")
        args["sampled_reads"] = n_total
        if head[1] >= 0:
            mc._bases_cache[tuple(args["seqfiles"])] = head[1]
        block, lo = scatter_reads(reads, n_total, L, rank, world, device=dev)
        model = mc._model()
        fams = model["families"]
        eng = mc._engine(device)
        eng.set_run(L, model["pars"][str(L)], fams)
        if nccl and world > 1:                                                 # the block is already in this GPU's HBM
            n_mine = block.numel() // L
            eng.attach(block.data_ptr(), n_mine)
            parts = []
            for off in range(0, n_mine, 2000000):
                eng.run_range(off, min(2000000, n_mine - off), first_read_id=lo + off)
                parts.append(eng.best_hits())
            best = np.concatenate(parts) if parts else np.zeros(0, _native.BEST_DTYPE)
        else:
            eng.lib.mc_set_keep_rows(eng.h, 0)
            try:
                eng.search(block, first_read_id=lo)
            finally:
                eng.lib.mc_set_keep_rows(eng.h, 1)
            best = eng.best_hits()
        acc = family_accumulators(best, len(fams))
        acc = all_reduce_accumulators(*acc, device=dev)
        agg = aggregate_from_accumulators(*acc, fams, mc.find_opt_pars(None, L))
        if not agg:
            raise SystemExit("\nError: No hits to marker proteins - cannot estimate genome size! Rerun program with more reads.")
        est = mc.estimate_average_genome_size(args, paths, agg)
        return est, args
    finally:
        mc.clean_up(paths)
