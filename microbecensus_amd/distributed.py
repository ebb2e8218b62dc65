"""Multi-GPU layer: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on MI355X).

The path shards embarrassingly: the accepted-read stream (after the sequential sampling / QC / duplicate
semantics of process_seqfile, which stay on the host) is cut into contiguous blocks, one per rank, the
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


def run_pipeline_distributed(args, device=None):
    """run_pipeline() over all ranks of the initialised torch.distributed group (one process per GPU; backend "nccl" = RCCL
    on MI355X, or gloo).  Every rank runs the (sequential, deterministic) sampler on the same input, searches its own
    contiguous block of the accepted reads on its GPU with global read ids, the per-family integer accumulators are summed
    with ONE all_reduce, and every rank finishes the estimate from the same sums.  Returns (est_ags, args) like run_pipeline;
    hits are integers and the 'cov' sums are finished from exact integer sums, so the result does not depend on the number
    of ranks (<= 1e-12 relative against the single-process sum order)."""
    import os
    import torch
    import torch.distributed as dist
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
    try:
        mc.process_seqfile(args, paths)
        reads = mc._run_cache[paths["tempfile"]]["reads"]
        L = args["read_length"]
        lo, hi = shard_bounds(len(reads), rank, world)
        model = mc._model()
        fams = model["families"]
        eng = mc._engine(device)
        eng.set_run(L, model["pars"][str(L)], fams)
        eng.search(reads[lo:hi], first_read_id=lo)
        best = eng.best_hits()
        acc = family_accumulators(best, len(fams))
        dev = torch.device("cuda", device) if (dist.is_initialized() and dist.get_backend() == "nccl") else None
        acc = all_reduce_accumulators(*acc, device=dev)
        agg = aggregate_from_accumulators(*acc, fams, mc.find_opt_pars(None, L))
        if not agg:
            raise SystemExit("\nError: No hits to marker proteins - cannot estimate genome size! Rerun program with more reads.")
        est = mc.estimate_average_genome_size(args, paths, agg)
        return est, args
    finally:
        mc.clean_up(paths)
