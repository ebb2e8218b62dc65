"""Multi-GPU layer: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on MI355X).

The path shards embarrassingly: the accepted-read stream (after the sequential sampling / QC / duplicate
semantics of process_seqfile, which stay on the host of rank 0) is dealt in batches of 2 M reads to the ranks
while it is being sampled (run_pipeline_distributed; bench.py cuts a resident stream into contiguous blocks instead), the
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


STREAM_BATCH = 2000000      # accepted reads per dealt batch (MC_DIST_BATCH in the environment: tests deal small batches)


def _stream_batch():
    import os
    try:
        v = int(os.environ.get("MC_DIST_BATCH", "0"))
    except ValueError:
        v = 0
    return v if 1000 <= v <= STREAM_BATCH else STREAM_BATCH


class _Dealer:
    """Rank 0's side of the streamed multi-GPU pipeline: the native sampler runs on its own thread (mc_reader_start); this
    thread fetches batches of accepted reads as they appear (mc_reader_fetch, straight into pinned memory) and deals batch k to
    rank k mod world - over RCCL from a GPU staging buffer, over gloo from host memory; the batches of rank 0 itself go into a
    local queue.  Nothing waits for the whole library: sampling, dealing and searching overlap, and no rank ever holds more than
    two batches.  A header (reads in the batch, index of its first read) precedes every batch; (0, n_total) ends the stream,
    (-1, 0) reports a sampler error."""

    def __init__(self, reader, read_len, world, nccl, dev, local_q):
        import threading
        self.rd, self.L, self.world, self.nccl, self.dev, self.q = reader, read_len, world, nccl, dev, local_q
        self.error, self.n_total = None, 0
        self.trace = []                                            # (what, batch, t0, t1): the test of the overlap reads it
        self.th = threading.Thread(target=self._run, daemon=True)

    def start(self):
        self.th.start()

    def join(self):
        self.th.join()

    def _run(self):
        import ctypes as C
        import threading
        import time
        import torch
        import torch.distributed as dist
        lib, r, L, B = self.rd.lib, self.rd.r, self.L, _stream_batch()
        pin = torch.cuda.is_available()
        bufs = [torch.empty(B * L, dtype=torch.uint8, pin_memory=pin) for _ in range(2)]
        stage = [torch.empty(B * L, dtype=torch.uint8, device=self.dev) for _ in range(2)] if self.nccl else None
        pending = [None, None]                                     # the send that last used buffer i
        hdr_dev = self.dev if self.nccl else torch.device("cpu")
        try:
            if lib.mc_reader_start(r) != 0:
                raise RuntimeError(lib.mc_reader_last_error().decode())
            at, k = 0, 0
            while True:
                i = k % 2
                if pending[i] is not None:
                    for w in pending[i]:
                        w.wait()
                    if self.nccl:
                        torch.cuda.current_stream().synchronize()
                    pending[i] = None
                t0 = time.time()
                n = lib.mc_reader_fetch(r, at, B, C.c_void_p(bufs[i].data_ptr()))
                t1 = time.time()
                self.trace.append(("fetch", k, t0, t1))
                if n < 0:
                    raise _SamplerError(lib.mc_reader_last_error().decode(), n)
                if n == 0:
                    break
                dst = k % self.world
                if dst == 0:                                       # rank 0's own batch: searched from the pinned buffer, which comes back with the event
                    done = threading.Event()
                    self.q.put((bufs[i][: n * L].numpy().reshape(n, L), at, done))
                    pending[i] = [done]
                else:
                    hdr = torch.tensor([n, at], dtype=torch.int64, device=hdr_dev)
                    payload = bufs[i][: n * L]
                    if self.nccl:
                        stage[i][: n * L].copy_(payload, non_blocking=True)
                        payload = stage[i][: n * L]
                    pending[i] = [dist.isend(hdr, dst), dist.isend(payload, dst)]
                    pending[i].append(_Keep(hdr))
                self.trace.append(("deal", k, t1, time.time()))
                at += n
                k += 1
            for p in pending:
                if p is not None:
                    for w in p:
                        w.wait()
            total = lib.mc_reader_join(r)
            if total < 0:
                raise _SamplerError(lib.mc_reader_last_error().decode(), total)
            self.n_total = int(total)
            end = [int(0), self.n_total]
        except _SamplerError as e:
            self.error = e
            end = [-1, 0]
        except BaseException as e:                                  # noqa: BLE001 - reported by the main thread
            self.error = e
            end = [-1, 0]
        try:
            import torch.distributed as dist2
            for dst in range(1, self.world):
                dist2.send(torch.tensor(end, dtype=torch.int64, device=hdr_dev), dst)
        finally:
            self.q.put(None)


class _Keep:
    """keeps a tensor alive until its asynchronous send has been waited for"""

    def __init__(self, t):
        self.t = t

    def wait(self):
        self.t = None


class _SamplerError(Exception):
    def __init__(self, msg, code):
        Exception.__init__(self, msg)
        self.code = code


def _receive_batches(read_len, nccl, dev, q):
    """A rank > 0: receives (header, batch) pairs from rank 0 into two buffers in turn and queues them for the searching thread;
    returns the total number of sampled reads (the end header carries it), or raises what rank 0 reported."""
    import torch
    import torch.distributed as dist
    B, L = _stream_batch(), read_len
    rdev = dev if nccl else torch.device("cpu")
    bufs = [torch.empty(B * L, dtype=torch.uint8, device=rdev) for _ in range(2)]
    free = [None, None]                                             # events the searching thread sets when it is done with buffer i
    import threading
    k = 0
    while True:
        hdr = torch.empty(2, dtype=torch.int64, device=rdev)
        dist.recv(hdr, 0)
        n, first = [int(v) for v in hdr.cpu().tolist()]
        if n <= 0:
            q.put(None)
            if n < 0:
                raise Exception("the sampler on rank 0 failed")
            return first
        i = k % 2
        if free[i] is not None:
            free[i].wait()
        dist.recv(bufs[i][: n * L], 0)
        if nccl:
            torch.cuda.current_stream().synchronize()               # the engine launches on its own streams: the batch must have landed
        free[i] = threading.Event()
        q.put((bufs[i][: n * L], first, free[i]))
        k += 1


def run_pipeline_distributed(args, device=None):
    """run_pipeline() over all ranks of the initialised torch.distributed group (one process per GPU; backend "nccl" = RCCL
    on MI355X, or gloo), STREAMED: rank 0 runs the (sequential, deterministic) sampler on a thread of its own and deals
    batches of 2 M accepted reads round-robin to the ranks as they appear (_Dealer; RCCL: pinned host -> GPU staging -> xGMI);
    every rank searches the batches it is dealt with global read ids while the next one arrives; the per-family integer
    accumulators are summed with ONE all_reduce at the end, and every rank finishes the estimate from the same sums.  Peak
    memory per rank is two batches, whatever the size of the library.  Returns (est_ags, args) like run_pipeline; hits are
    integers and the 'cov' sums are finished from exact integer sums, so the result does not depend on the number of ranks
    (<= 1e-12 relative against the single-process sum order)."""
    import os
    import queue
    import sys
    import threading
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
    mc._cap_host_threads(args.get("threads"))
    mc.impute_missing_args(args)
    mc.check_arguments(args)
    args["verbose"] = bool(args.get("verbose")) and rank == 0
    nccl = dist.is_initialized() and dist.get_backend() == "nccl"
    dev = torch.device("cuda", device) if nccl else None
    if nccl:
        torch.cuda.set_device(device)
    try:
        L = args["read_length"]
        model = mc._model()
        fams = model["families"]
        eng = mc._engine(device)
        eng.set_run(L, model["pars"][str(L)], fams)
        q = queue.Queue(maxsize=1 if rank else 2)
        parts, search_err = [], []

        def search_loop():                                                     # this rank's batches, as they arrive
            try:
                eng.lib.mc_set_keep_rows(eng.h, 0)
                while True:
                    item = q.get()
                    if item is None:
                        return
                    if search_err:
                        if len(item) > 2:
                            item[2].set()
                        continue                                               # (keep draining so that the dealer / receiver never blocks)
                    try:
                        if rank == 0 or not nccl:
                            blk = item[0] if rank == 0 else item[0].numpy().reshape(-1, L)             # (host memory: rank 0's own batch, or gloo)
                            eng.search(blk, first_read_id=item[1])
                        else:                                                  # the batch is already in this GPU's HBM
                            n = item[0].numel() // L
                            eng.attach(item[0].data_ptr(), n)
                            eng.run_range(0, n, first_read_id=item[1])
                            eng.attach(0, 0)
                        parts.append(eng.best_hits())
                    except BaseException as e:                                 # noqa: BLE001
                        search_err.append(e)
                    finally:
                        if len(item) > 2:
                            item[2].set()
            finally:
                eng.lib.mc_set_keep_rows(eng.h, 1)

        worker = threading.Thread(target=search_loop, daemon=True)
        worker.start()
        head = [0, -1, ""]
        rd = None
        try:
            if rank == 0:
                rd = _native.Reader(args["seqfiles"], L, args["nreads"], args["file_type"] == "fastq", args.get("quality_offset") or 0,
                                    args["min_quality"], args["mean_quality"], args["max_unknown"], args["filter_dups"])
                dealer = _Dealer(rd, L, world, nccl, dev, q)
                dealer.start()
                dealer.join()
                worker.join()
                run_pipeline_distributed.last_trace = dealer.trace
                if dealer.error is not None:
                    head = [-1, -1, str(dealer.error)] if isinstance(dealer.error, _SamplerError) and dealer.error.code == -3 else [-2, -1, str(dealer.error)]
                else:
                    st = rd.stats()
                    head = [dealer.n_total, int(st["bases"]) if st.get("exhausted") else -1, ""]
            else:
                try:
                    _receive_batches(L, nccl, dev, q)
                except Exception:
                    pass                                                       # rank 0 broadcasts what happened
                worker.join()
        finally:
            if rd is not None:
                rd.close()
        if world > 1:
            dist.broadcast_object_list(head, src=0)
        if head[0] == -1:
            raise Exception(head[2])                                           # the reference raises inside run_pipeline
        if head[0] < 0:
            raise RuntimeError(head[2])
        if search_err:
            raise search_err[0]
        n_total = head[0]
        if n_total == 0:
            sys.exit("\nError! No reads remaining after filtering!")
        args["sampled_reads"] = n_total
        if head[1] >= 0:
            mc._bases_cache[tuple(args["seqfiles"])] = head[1]
        best = np.concatenate(parts) if parts else np.zeros(0, _native.BEST_DTYPE)
        acc = family_accumulators(best, len(fams))
        acc = all_reduce_accumulators(*acc, device=dev)
        agg = aggregate_from_accumulators(*acc, fams, mc.find_opt_pars(None, L))
        if not agg:
            raise SystemExit("\nError: No hits to marker proteins - cannot estimate genome size! Rerun program with more reads.")
        est = mc.estimate_average_genome_size(args, paths, agg)
        return est, args
    finally:
        mc.clean_up(paths)
