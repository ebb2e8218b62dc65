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


def _usable_cores():
    """CPUs this process may use: the cgroup quota, else the affinity mask, else os.cpu_count()."""
    import os
    n = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = max(1, int(round(float(q) / float(per))))
    except Exception:                                               # noqa: BLE001
        pass
    try:
        a = len(os.sched_getaffinity(0))
        n = a if n is None else min(n, a)
    except Exception:                                               # noqa: BLE001
        pass
    return n or os.cpu_count() or 1


STREAM_BATCH = 2000000      # accepted reads per dealt batch (MC_DIST_BATCH in the environment: tests deal small batches)


def _stream_batch():
    import os
    try:
        v = int(os.environ.get("MC_DIST_BATCH", "0"))
    except ValueError:
        v = 0
    return v if 1000 <= v <= STREAM_BATCH else STREAM_BATCH


_stream_seq = [0]


def _control_store():
    """The key-value store of the process group (a TCPStore), under a prefix of this stream's own: the small control messages -
    batch headers from rank 0, credits from the ranks - travel through it, beside the payloads (RCCL or gloo send / recv).  The
    store's get blocks until a key exists and its add is an atomic counter: exactly what headers and credits need, from any thread,
    with nothing posted on the process group that could queue behind (or deadlock with) a GPU transfer.  (gloo's irecv cannot be
    polled - is_completed() stays false until wait() - so credits cannot be gathered from several ranks with point-to-point messages.)
    Every rank calls this at the same point of the run."""
    import torch.distributed as dist
    from torch.distributed import distributed_c10d as c10d
    _stream_seq[0] += 1
    return dist.PrefixStore("mcensus_stream_%d" % _stream_seq[0], c10d._get_default_store())


def _store_wait_get(store, key):
    """store.get that waits as long as it takes (a sampler may take minutes between batches; the store's own time-out raises)"""
    import datetime
    while True:
        try:
            store.wait([key], datetime.timedelta(seconds=30))
            return store.get(key)
        except Exception as e:                                      # noqa: BLE001 - a time-out: ask again
            if "imeout" not in str(e) and "imed out" not in str(e):
                raise


def _store_poll_get(store, key):
    """The same for a key that several threads of a rank may wait for while others set theirs (the hand-overs of the .gz slices: two rounds
    are under way at a time): a TCPStore client serves one call at a time, so a thread blocked in wait() would hold up the set() of the
    thread beside it - whose key the next rank is waiting for - until its time-out.  check() returns at once."""
    import time
    nap = 0.0002
    while not store.check([key]):
        time.sleep(nap)
        nap = min(0.002, nap * 1.2)
    return store.get(key)


class _Dealer:
    """Rank 0's side of the streamed multi-GPU pipeline: the native sampler runs on its own thread (mc_reader_start); this
    thread fetches batches of accepted reads as they appear (mc_reader_fetch, straight into pinned memory) and deals every
    batch to the next rank that is FREE - over RCCL from a GPU staging buffer, over gloo from host memory; the batches of rank 0
    itself go into a local queue.  A rank is free when it holds a credit: every rank starts with two (its two buffers) and
    returns one whenever it has finished a batch - a slow rank (a busy GPU, a throttled one) simply asks less often instead of
    holding up the stream, which dealing round robin (round 3) did.  Nothing waits for the whole library: sampling, dealing and
    searching overlap, and no rank ever holds more than two batches.  A header (reads in the batch, index of its first read)
    precedes every batch; (0, n_total) ends the stream, (-1, 0) reports a sampler error.  Headers and credits go through the
    process group's store (_control_store), payloads over the group itself."""

    def __init__(self, reader, read_len, world, nccl, dev, local_q, store):
        import threading
        self.rd, self.L, self.world, self.nccl, self.dev, self.q, self.store = reader, read_len, world, nccl, dev, local_q, store
        self.error, self.n_total = None, 0
        self.trace = []                                            # (what, batch, t0, t1[, dst]): the tests read it
        self.local_credit = 2
        self.lock = threading.Lock()
        self.th = threading.Thread(target=self._run, daemon=True)

    def start(self):
        self.th.start()

    def join(self):
        self.th.join()

    def local_done(self):
        with self.lock:
            self.local_credit += 1

    @staticmethod
    def _check_peers(store, world):
        """A rank whose receiver failed (a broken recv, a store error) says so under the key x<rank>: the stream ends for
        everybody - that rank returns no more credits, and a batch dealt to it would never be matched."""
        for w in range(1, world):
            try:
                dead = store.check(["x%d" % w])
            except Exception:                                       # noqa: BLE001 - a store without check(): nothing to look at
                return
            if dead:
                raise RuntimeError("rank %d failed while receiving its batches" % w)

    def _wait_buffer(self, works):
        """Waits until nothing uses a staging buffer any more - and keeps looking at the peers while it does: a receiver
        that dies with one of its batches in flight leaves an isend nobody will ever match, and neither backend lets one ask whether
        a send is done without waiting for it (gloo's is_completed() stays false until wait(); RCCL's wait() is a stream dependency
        that the synchronize behind it turns into a host wait).  So the wait itself runs on a helper thread and this one checks the
        peers once a second; if one has failed the helper is left behind (daemon) and the stream ends for everybody."""
        import threading
        done, err = threading.Event(), []

        def waiter():
            try:
                import torch
                for w in works:
                    w.wait()
                if self.nccl:
                    torch.cuda.set_device(self.dev)
                    torch.cuda.current_stream(self.dev).synchronize()
            except BaseException as e:                              # noqa: BLE001 - raised by the dealer's thread below
                err.append(e)
            finally:
                done.set()
        threading.Thread(target=waiter, daemon=True).start()
        while not done.wait(1.0):
            self._check_peers(self.store, self.world)
        if err:
            raise err[0]

    def _run(self):
        import ctypes as C
        import threading
        import time
        import torch
        import torch.distributed as dist
        lib, r, L, B = self.rd.lib, self.rd.r, self.L, _stream_batch()
        if self.nccl:
            torch.cuda.set_device(self.dev)                        # (the current device is per thread: the staging copies and their stream belong to this rank's GPU)
        pin = torch.cuda.is_available()
        NB = 3
        bufs = [torch.empty(B * L, dtype=torch.uint8, pin_memory=pin) for _ in range(NB)]
        stage = [torch.empty(B * L, dtype=torch.uint8, device=self.dev) for _ in range(NB)] if self.nccl else None
        pending = [None] * NB                                      # what still uses buffer i
        world, store = self.world, self.store
        sent = [0] * world                                         # batches dealt to rank w (= headers written for it)
        last = world - 1
        try:
            if lib.mc_reader_start(r) != 0:
                raise RuntimeError(lib.mc_reader_last_error().decode())
            at, k = 0, 0
            while True:
                i = k % NB
                self._check_peers(store, world)
                if pending[i] is not None:
                    self._wait_buffer(pending[i])
                    pending[i] = None
                t0 = time.time()
                n = lib.mc_reader_fetch(r, at, B, C.c_void_p(bufs[i].data_ptr()))
                t1 = time.time()
                self.trace.append(("fetch", k, t0, t1))
                if n < 0:
                    raise _SamplerError(lib.mc_reader_last_error().decode(), n)
                if n == 0:
                    break
                dst, polls = -1, 0
                while dst < 0:                                      # the next free rank, in turn
                    for j in range(1, world + 1):
                        w = (last + j) % world
                        if w == 0:
                            with self.lock:
                                if self.local_credit > 0:
                                    self.local_credit -= 1; dst = 0
                        elif store.add("c%d" % w, 0) > sent[w]:    # (credits the rank has given so far: 2 + the batches it has finished)
                            dst = w
                        if dst >= 0:
                            break
                    if dst < 0:
                        time.sleep(0.0005)
                        polls += 1
                        if polls % 2000 == 0:                       # (once a second while nobody is free)
                            self._check_peers(store, world)
                last = dst
                if dst == 0:                                       # rank 0's own batch: searched from the pinned buffer, which comes back with the event
                    done = threading.Event()
                    self.q.put((bufs[i][: n * L].numpy().reshape(n, L), at, done))
                    pending[i] = [done]
                else:
                    payload = bufs[i][: n * L]
                    if self.nccl:
                        stage[i][: n * L].copy_(payload, non_blocking=True)
                        payload = stage[i][: n * L]
                    # The payload is posted BEFORE its header is written: a rank that has read a header always finds its batch on
                    # the way, and if staging or isend fails no header exists - the end marker below lands on the very key the
                    # rank is waiting for (header first, then a failure, left the receiver in recv for ever).
                    pending[i] = [dist.isend(payload, dst)]
                    store.set("h%d/%d" % (dst, sent[dst]), "%d,%d" % (n, at))
                    sent[dst] += 1
                self.trace.append(("deal", k, t1, time.time(), dst))
                at += n
                k += 1
            for p in pending:
                if p is not None:
                    self._wait_buffer(p)
            total = lib.mc_reader_join(r)
            if total < 0:
                raise _SamplerError(lib.mc_reader_last_error().decode(), total)
            self.n_total = int(total)
            end = (0, self.n_total)
        except _SamplerError as e:
            self.error = e
            end = (-1, 0)
        except BaseException as e:                                  # noqa: BLE001 - reported by the main thread
            self.error = e
            end = (-1, 0)
        try:
            for dst in range(1, world):
                store.set("h%d/%d" % (dst, sent[dst]), "%d,%d" % end)
        finally:
            self.q.put(None)


class _SamplerError(Exception):
    def __init__(self, msg, code):
        Exception.__init__(self, msg)
        self.code = code


def _receive_batches(read_len, nccl, dev, q, store, rank):
    """A rank > 0: gives rank 0 two credits (its two buffers; the searching thread gives one more whenever it has finished with a
    buffer), waits for the headers rank 0 writes for it, receives every batch into a free buffer and queues it for the searching
    thread; returns the total number of sampled reads (the end header carries it), or raises what rank 0 reported."""
    import threading
    import torch
    import torch.distributed as dist
    B, L = _stream_batch(), read_len
    rdev = dev if nccl else torch.device("cpu")
    if nccl:
        torch.cuda.set_device(dev)
    bufs = [torch.empty(B * L, dtype=torch.uint8, device=rdev) for _ in range(2)]
    busy = [None, None]                                             # the event the searching thread sets when it is done with buffer i
    store.add("c%d" % rank, 2)
    j = 0
    try:
        while True:
            key = "h%d/%d" % (rank, j)
            n, first = [int(v) for v in _store_wait_get(store, key).decode().split(",")]
            try:
                store.delete_key(key)                               # (consumed: the store does not grow with the library)
            except Exception:                                       # noqa: BLE001 - a store without delete_key (FileStore of old versions)
                pass
            j += 1
            if n <= 0:
                if n < 0:
                    raise Exception("the sampler on rank 0 failed")
                return first
            i = 0 if (busy[0] is None or busy[0].is_set()) else 1   # (rank 0 deals only against credits: a buffer is free)
            dist.recv(bufs[i][: n * L], 0)
            if nccl:
                torch.cuda.current_stream(dev).synchronize()        # the engine launches on its own streams: the batch must have landed
            busy[i] = threading.Event()
            q.put((bufs[i][: n * L], first, busy[i]))
    finally:
        q.put(None)                                                 # whatever happened here, the searching thread ends (it waited in q.get() for ever)


def stream_batches(reader, read_len, on_batch, device=None):
    """The streamed dealing by itself: rank 0 samples with `reader` (None on the other ranks) and deals batches to whoever is
    free; on every rank on_batch(block, first_read_id) is called for the batches it gets - block: (n, read_len) uint8 numpy array
    (host memory), or with RCCL on ranks > 0 a 1-D uint8 torch tensor in this rank's HBM.  Returns (n_total, dealer trace or None,
    error or None) - the error already agreed on by all ranks.  run_pipeline_distributed searches the batches; the tests check
    the dealing with plain checksums (no GPU needed)."""
    import os
    import queue
    import threading
    import time
    import torch
    import torch.distributed as dist
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    nccl = dist.is_initialized() and dist.get_backend() == "nccl"
    dev = torch.device("cuda", device) if nccl else None
    store = _control_store() if world > 1 else None
    if world > 1:
        # The pairs rank 0 <-> rank w meet ONCE before any header logic: on an RCCL group that was initialised lazily
        # (init_process_group without device_id) the first point-to-point call of a pair creates its communicator and blocks until the
        # peer enters the matching call - the dealer posts a batch BEFORE it writes the header the receiver waits for, so the first
        # batch of every rank would wait for a recv that waits for a header.  Here both sides enter unconditionally.
        tiny = torch.zeros(1, dtype=torch.uint8, device=dev if nccl else torch.device("cpu"))
        if rank == 0:
            for w in range(1, world):
                dist.send(tiny, w)
        else:
            dist.recv(tiny, 0)
        if nccl:
            torch.cuda.current_stream(dev).synchronize()
    q = queue.Queue(maxsize=2)
    errs = []
    slow = os.environ.get("MC_DIST_SLOW", "")                      # "<rank>:<milliseconds>": that rank dawdles after every batch (tests)
    slow_s = float(slow.split(":")[1]) / 1e3 if slow and int(slow.split(":")[0]) == rank else 0.0
    dealer = None

    def loop():
        while True:
            item = q.get()
            if item is None:
                return
            try:
                if not errs:
                    on_batch(item[0], item[1])
                    if slow_s:
                        time.sleep(slow_s)
            except BaseException as e:                             # noqa: BLE001
                errs.append(e)
            finally:                                               # (keep draining so that the dealer / receiver never blocks)
                item[2].set()
                if dealer is not None:
                    dealer.local_done()
                elif store is not None:
                    store.add("c%d" % rank, 1)                     # a credit: this rank can take another batch

    worker = threading.Thread(target=loop, daemon=True)
    n_total, trace, status = 0, None, [0, -1, ""]
    if rank == 0:
        dealer = _Dealer(reader, read_len, world, nccl, dev, q, store)
        worker.start()
        dealer.start()
        dealer.join()
        worker.join()
        trace = dealer.trace
        if dealer.error is not None:
            status = [-1, -1, str(dealer.error)] if isinstance(dealer.error, _SamplerError) and dealer.error.code == -3 else [-2, -1, str(dealer.error)]
        else:
            status = [dealer.n_total, 0, ""]
    else:
        worker.start()
        try:
            _receive_batches(read_len, nccl, dev, q, store, rank)
        except Exception as e:                                     # noqa: BLE001 - "the sampler on rank 0 failed": rank 0 broadcasts what happened;
            if "sampler on rank 0" not in str(e):                  # anything else (a failed recv, a store error) is THIS rank's error and travels
                errs.append(e)                                     # with the flags below, so that all ranks raise together; rank 0 stops dealing
                try:
                    store.set("x%d" % rank, "1")
                except Exception:                                  # noqa: BLE001
                    pass
        worker.join()
    # every rank learns how the stream ended AND whether any rank failed while searching: all raise together (a rank that raised
    # alone left the others waiting in the all_reduce)
    if world > 1:
        dist.broadcast_object_list(status, src=0)
        flags = [None] * world
        dist.all_gather_object(flags, str(errs[0]) if errs else "")
        bad = [f for f in flags if f]
    else:
        bad = [str(errs[0])] if errs else []
    err = None
    if status[0] == -1:
        err = Exception(status[2])                                 # the reference raises inside run_pipeline
    elif status[0] < 0:
        err = RuntimeError(status[2])
    elif bad:
        err = errs[0] if errs else RuntimeError("a rank failed while searching: " + bad[0])
    n_total = max(0, status[0])
    return n_total, trace, err


SLICE_BYTES = 256 << 20     # bytes of a file one rank samples per round (MC_DIST_SLICE in the environment: tests cut small files into many slices)


def sharded_sampling_usable(args):
    """Can every rank sample its own slices of the input?  Without -d (with it: stream_batches_sharded_dups), on plain regular files
    (byte windows), on .bz2 files all of whose streams check out (block ranges: the blocks of a bzip2 file are independent,
    csrc/mc_pbzip2.h) and on .gz files the parallel reader takes (chunk ranges: a member cannot be entered in the middle, but it can be
    decoded from the middle speculatively - the ranks hand the 32 KB windows along, csrc/mc_pgzip.h start_slice), and unless
    MC_DIST_SHARDED=0."""
    import os
    from . import _native
    if os.environ.get("MC_DIST_SHARDED") == "0" or args.get("filter_dups"):
        return False
    for p in args["seqfiles"]:
        if not os.path.isfile(p):
            return False
        if p.endswith(".gz") and (os.environ.get("MC_DIST_GZ") == "0" or _native.gz_chunks(p, _gz_chunk_bytes()) <= 0):
            return False
        if p.endswith(".bz2") and _native.bz2_blocks(p) <= 0:
            return False
    return True


def _gz_chunk_bytes():
    """compressed bytes per chunk of a .gz file decoded across the ranks (MC_DIST_GZ_CHUNK: tests cut small files into many)"""
    import os
    try:
        v = int(os.environ.get("MC_DIST_GZ_CHUNK", "0"))
    except ValueError:
        v = 0
    return v if v >= 4096 else (1 << 20)


def stream_batches_sharded(args, on_batch, device=None):
    """process_seqfile (reference microbe_census.py:328-367) with a sampler on EVERY rank.  One sampler on rank 0 delivers 60 M
    reads/s of plain FASTQ (9 M/s of .gz) where eight MI355X search 400 M reads/s: file -> AGS could not scale past 1.3 GPUs.
    The files are walked in rounds of `world` slices of SLICE_BYTES; rank r samples slice r of the round with the native reader
    on that byte window (mc_reader_open_range: windows cut a file into whole records by one rule, whoever reads them); the
    ranks exchange their counts of accepted reads; the prefix sum gives every rank the global index of its first read - and the
    point where the head-take ends: a rank keeps the first nreads - prefix of its reads, the ranks behind it none, and the rounds
    stop.  on_batch(block, first_read_id) gets the kept reads ((n, L) uint8, host memory).  The sampling of the next round runs
    beside the search of this one.
    Returns (n_total, stats, bases, status): stats = the reference's counters (records met before the head-take ended), bases =
    count_bases() when every file was read to its end (else -1), status 0, or 1 = a window did not end on a record boundary
    (multi-line FASTQ whose qualities look like headers): the caller falls back to the sampler on rank 0; 2 = a rank failed
    (its sampler raised, or on_batch - the search - did): stats then carries the agreed message under "error" and the caller
    raises on every rank instead of searching everything a second time."""
    import os
    import sys
    import threading
    import time
    import numpy as np
    import torch
    import torch.distributed as dist
    from . import _native
    rank, world = dist.get_rank(), dist.get_world_size()
    nccl = dist.get_backend() == "nccl"
    tdev = torch.device("cuda", device) if nccl else torch.device("cpu")
    L, fastq = args["read_length"], args["file_type"] == "fastq"
    qoff = args.get("quality_offset") or 0
    nreads = args["nreads"] if args["nreads"] is not None else (1 << 62)
    try:
        S = int(os.environ.get("MC_DIST_SLICE", "0")) or SLICE_BYTES
    except ValueError:
        S = SLICE_BYTES
    # a round = `world` consecutive slices of one file: S bytes each of a plain file, KB blocks each of a .bz2 file, KC chunks each of a .gz
    # file (the same list on every rank)
    KB = max(1, S // 900000)
    GZC = _gz_chunk_bytes()
    KC = max(1, min(32, S // GZC))
    kind = "@" if fastq else ">"
    rounds, gz_step = [], {}
    for fi, p in enumerate(args["seqfiles"]):
        if p.endswith(".bz2"):
            size = _native.bz2_blocks(p)
            rounds += [(p, base, size, KB, fi) for base in range(0, max(size, 1), KB * world)]
        elif p.endswith(".gz"):
            size = _native.gz_chunks(p, GZC)
            kc = max(min(4, KC), min(KC, -(-size // world)))        # (a file of fewer than `world` full slices: smaller ones, so that every rank has one)
            gz_step[fi] = kc
            rounds += [(p, base, size, kc, fi) for base in range(0, max(size, 1), kc * world)]
        else:
            size = os.path.getsize(p)
            rounds += [(p, base, size, S, fi) for base in range(0, max(size, 1), S * world)]
    gz_store = _control_store() if any(p.endswith(".gz") for p in args["seqfiles"]) else None
    trace_on, t_zero = bool(os.environ.get("MC_DIST_TRACE")), time.time()

    def run_slice(p, lo, hi, cap, fi=0, state_in=None, publish=True, on_published=None):
        """(reader, accepted reads, state of the slice in front - .gz only) for slice [lo, hi) of file p"""
        if p.endswith(".gz"):
            # the chain of a .gz file: slice i learns from slice i - 1 where that ended (and the 32 KB in front of it), tells slice i + 1 the same
            # as soon as it knows - through the process group's store, from whichever thread -, and the members' CRCs follow the same way
            i = lo // gz_step[fi]
            rd = _native.Reader.on_gz_part(p, lo, hi, GZC, kind, L, max(1, cap), fastq, qoff, args["min_quality"], args["mean_quality"], args["max_unknown"])
            n, done = 0, False
            tr = [time.time()] if trace_on else None
            try:
                rd.start()
                if lo > 0:
                    if state_in is None:
                        state_in = _store_poll_get(gz_store, "s%d/%d" % (fi, i - 1))
                    rd.gz_provide(state_in)
                if tr: tr.append(time.time())
                end = rd.gz_end_state()
                if publish:
                    gz_store.set("s%d/%d" % (fi, i), end or b"")
                if tr: tr.append(time.time())
                if on_published is not None:                       # the slice is decoded, its text is being sampled: the next round's decoding may start beside that
                    on_published()
                n = rd.join()
                if tr: tr.append(time.time())
                if publish:
                    crc_in = _store_poll_get(gz_store, "c%d/%d" % (fi, i - 1)) if lo > 0 else bytes(12)
                    if len(crc_in) != 12:
                        raise RuntimeError("the slice in front of this one failed")
                    gz_store.set("c%d/%d" % (fi, i), rd.gz_finish(crc_in))
                done = True
                if tr:                                             # MC_DIST_TRACE: when the slice was opened, had its predecessor's state, was decoded, sampled, checked
                    sys.stderr.write("gz slice %d rank %d: open %.3f state_in +%.3f decoded +%.3f sampled +%.3f crc +%.3f (%d reads) %s\n" % (i, rank, tr[0] - t_zero, tr[1] - tr[0], tr[2] - tr[1], tr[3] - tr[2], time.time() - tr[3], n, rd.times()))
            finally:
                if not done and publish:                               # whoever waits for this slice must not wait for ever
                    for k in ("s%d/%d" % (fi, i), "c%d/%d" % (fi, i)):
                        try:
                            if not gz_store.check([k]):
                                gz_store.set(k, b"")
                        except Exception:                           # noqa: BLE001
                            pass
            return rd, n, state_in
        if p.endswith(".bz2"):
            rd = _native.Reader.on_bz2_part(p, lo, hi, kind, L, max(1, cap), fastq, qoff, args["min_quality"], args["mean_quality"], args["max_unknown"])
        else:
            rd = _native.Reader.on_range(p, lo, hi, L, max(1, cap), fastq, qoff, args["min_quality"], args["mean_quality"], args["max_unknown"])
        return rd, rd.run(), None

    lock = threading.Lock()
    started, published, at_round, stopped = {}, set(), [0], [False]

    def ensure(j):
        with lock:
            if j < len(rounds) and j not in started and not stopped[0]:
                started[j] = sample(j, nreads - total)             # (an upper bound of what is still wanted: total only grows)

    def decoded(j):
        """round j's slice of this rank is decoded (.gz): round j + 1 starts now - its decoding beside the sampling of round j and the
        search of round j - 1 - unless the main loop is further behind than that (no rank starts a round more than two in front of the
        one its main loop is at; when the loop ends, the slices of those rounds that this rank never started are published as failed:
        whoever waits for a hand-over gets an answer)"""
        with lock:
            published.add(j)
            go = at_round[0] >= j - 1
        if go:
            ensure(j + 1)

    def sample(j, cap):
        p, base, size, step, fi = rounds[j]
        lo = min(size, base + rank * step)
        hi = min(size, lo + step)
        box = {"fi": fi}

        def work():
            try:
                if hi > lo:
                    try:
                        rd, n, st_in = run_slice(p, lo, hi, cap, fi, on_published=lambda: decoded(j))
                    finally:
                        pass
                    box["rd"], box["n"], box["state_in"] = rd, n, st_in
                    box["st"] = rd.stats()
                else:
                    box["n"], box["st"] = 0, {"too_short": 0, "low_qual": 0, "records": 0, "bases": 0, "ragged_end": 0, "exhausted": 1}
            except BaseException as e:                             # noqa: BLE001
                if p.endswith((".bz2", ".gz")) and (isinstance(e, RuntimeError) or p.endswith(".gz")):
                    # (no record start near a block / chunk boundary, a slice that does not decode, a CRC that does not match: the sampler on
                    # rank 0 decides - and reports what the reference would)
                    box["n"], box["st"] = 0, {"too_short": 0, "low_qual": 0, "records": 0, "bases": 0, "ragged_end": 1, "exhausted": 0}
                else:
                    box["err"] = e
        th = threading.Thread(target=work, daemon=True)
        th.start()
        return th, box, (p, lo, hi)

    total, status, cut = 0, 0, False
    stats = {"too_short": 0, "low_qual": 0, "dups": 0, "records": 0, "bases": 0}
    err = None
    ensure(0)
    last = -1
    for j in range(len(rounds)):
        with lock:
            at_round[0] = j
            go = [jj for jj in (j, j + 1) if jj in published]
        for jj in go:
            ensure(jj + 1)
        th, box, (p, lo, hi) = started[j]
        th.join()
        ensure(j + 1)
        last = j
        bad = 2 if "err" in box else 1 if (box.get("st") or {}).get("ragged_end") else 0
        n_acc = 0 if bad else int(box["n"])
        mine = torch.tensor([n_acc, bad], dtype=torch.int64, device=tdev)
        allc = [torch.zeros(2, dtype=torch.int64, device=tdev) for _ in range(world)]
        dist.all_gather(allc, mine)
        counts = [int(t[0].item()) for t in allc]
        worst = max(int(t[1].item()) for t in allc)
        if worst:
            status = worst                                         # 1: a ragged window (fall back); 2: a rank's sampler raised
            err = box.get("err")
        else:
            prefix = total + sum(counts[:rank])
            keep = max(0, min(n_acc, nreads - prefix))
            st = box["st"]
            if keep > 0 and prefix + n_acc >= nreads:              # the head-take ends in this slice: the counters stop with its last read
                rd2 = None
                try:
                    rd2, _, _ = run_slice(p, lo, hi, keep, box["fi"], state_in=box.get("state_in"), publish=False)
                    st = rd2.stats()
                finally:
                    if rd2 is not None:
                        rd2.close()
            if keep > 0 or prefix < nreads:                        # (a slice behind the end of the head-take was never looked at by the reference)
                for k in ("too_short", "low_qual", "records", "bases"):
                    stats[k] += int(st[k])
            if keep > 0:
                try:
                    on_batch(box["rd"].reads(n_acc)[:keep], prefix)
                except BaseException as e:                         # noqa: BLE001 - agreed on below
                    err = e
            total += sum(counts)
            cut = total >= nreads
        if "rd" in box:
            box["rd"].close()
        flag = torch.tensor([1 if err is not None else 0], dtype=torch.int64, device=tdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()):
            status = 2
        if status or cut:
            break
    with lock:
        stopped[0] = True
    for jj in range(last + 1, min(len(rounds), last + 4)):         # .gz rounds another rank may have started and this one never will: its slices count as failed
        p, base, size, step, fi = rounds[jj]
        if jj not in started and p.endswith(".gz") and min(size, base + rank * step) < size:
            for key in ("s%d/%d" % (fi, (base + rank * step) // step), "c%d/%d" % (fi, (base + rank * step) // step)):
                try:
                    if not gz_store.check([key]):
                        gz_store.set(key, b"")
                except Exception:                                   # noqa: BLE001
                    pass
    for jj in sorted(started):                                     # rounds sampled ahead and not needed
        if jj > last:
            started[jj][0].join()
            if "rd" in started[jj][1]:
                started[jj][1]["rd"].close()
    vec = torch.tensor([stats[k] for k in ("too_short", "low_qual", "records", "bases")], dtype=torch.int64, device=tdev)
    dist.all_reduce(vec)
    for k, v in zip(("too_short", "low_qual", "records", "bases"), vec.tolist()):
        stats[k] = int(v)
    n_total = min(total, nreads)
    bases = stats["bases"] if (status == 0 and not cut) else -1
    if status == 2:                                                # every rank learns what went wrong where (the first failing rank's message)
        msgs = [None] * world
        dist.all_gather_object(msgs, "" if err is None else "rank %d: %s: %s" % (rank, type(err).__name__, err))
        stats["error"] = next((m for m in msgs if m), "a rank failed")
    return n_total, stats, bases, status


def sharded_dups_usable(args):
    """-d with a sampler on every rank (stream_batches_sharded_dups): plain regular files, unless MC_DIST_SHARDED=0."""
    import os
    if os.environ.get("MC_DIST_SHARDED") == "0" or not args.get("filter_dups"):
        return False
    for p in args["seqfiles"]:
        if p.endswith((".gz", ".bz2")) or not os.path.isfile(p):
            return False
    return True


def stream_batches_sharded_dups(args, on_batch, device=None):
    """process_seqfile WITH -d (reference microbe_census.py:328-367; :345 the duplicate test comes before the quality filter, :354 only
    accepted reads enter the set) with a sampler on every rank.  The rule is class-local - a record's fate depends on nothing but the
    earlier records with its sequence or its reverse complement (csrc/mc_reader.cpp) - so what has to be seen in file order is 32 bytes
    per record, not the record: the files are walked in rounds of `world` slices as in stream_batches_sharded; rank r parses slice r,
    applies the quality filter and hashes every sequence and its reverse complement (mc_reader_describe); the ranks all_gather the
    descriptors of the round; EVERY rank walks them in file order through its own copy of the set (mc_dupset_walk: same input, same
    verdicts - sequences with equal hashes are compared on the file itself, which every rank maps) and so knows every record's
    verdict, the round's counters, where the head-take ends and the global index of its own first accepted read; it copies the accepted
    reads of its own slice (mc_reader_take) and searches them.  The next round is parsed beside the search of this one.
    Returns (n_total, stats, bases, status) as stream_batches_sharded; status 3 = the reference raises at a record the sampler reached
    (stats["error"] names the exception: every rank raises it)."""
    import os
    import threading
    import numpy as np
    import torch
    import torch.distributed as dist
    from . import _native
    rank, world = dist.get_rank(), dist.get_world_size()
    nccl = dist.get_backend() == "nccl"
    tdev = torch.device("cuda", device) if nccl else torch.device("cpu")
    L, fastq = args["read_length"], args["file_type"] == "fastq"
    qoff = args.get("quality_offset") or 0
    nreads = args["nreads"] if args["nreads"] is not None else (1 << 62)
    try:
        S = int(os.environ.get("MC_DIST_SLICE", "0")) or SLICE_BYTES
    except ValueError:
        S = SLICE_BYTES
    rounds = []
    for p in args["seqfiles"]:
        size = os.path.getsize(p)
        for base in range(0, max(size, 1), S * world):
            rounds.append((p, base, size))
    DT = _native.REC_DESC_DTYPE

    def sample(j):
        p, base, size = rounds[j]
        lo = min(size, base + rank * S)
        hi = min(size, lo + S)
        box = {}

        def work():
            try:
                rd = _native.Reader.on_range(p, lo, hi, L, 1 << 62, fastq, qoff, args["min_quality"], args["mean_quality"], args["max_unknown"])
                box["rd"] = rd
                box["d"] = rd.describe() if hi > lo else np.zeros(0, DT)
                box["ragged"] = bool(rd.stats()["ragged_end"]) if hi > lo else False
            except BaseException as e:                             # noqa: BLE001
                box["err"] = e
        th = threading.Thread(target=work, daemon=True)
        th.start()
        return th, box, p

    total, status, cut = 0, 0, False
    stats = {"too_short": 0, "low_qual": 0, "dups": 0, "records": 0, "bases": 0}
    err = None
    dupset = _native.DupSet()
    cur = sample(0) if rounds else None
    try:
        for j in range(len(rounds)):
            th, box, path = cur
            th.join()
            cur = sample(j + 1) if j + 1 < len(rounds) else None
            bad = 2 if "err" in box else 1 if box.get("ragged") else 0
            d = box.get("d") if not bad else None
            n_mine = 0 if d is None else len(d)
            mine = torch.tensor([n_mine, bad], dtype=torch.int64, device=tdev)
            allc = [torch.zeros(2, dtype=torch.int64, device=tdev) for _ in range(world)]
            dist.all_gather(allc, mine)
            counts = [int(t[0].item()) for t in allc]
            worst = max(int(t[1].item()) for t in allc)
            if worst:
                status = worst
                err = box.get("err")
            else:
                # the round's descriptors in file order: rank 0's slice, rank 1's, ...
                width = max(counts) * DT.itemsize
                buf = np.zeros(max(width, 1), np.uint8)
                if n_mine:
                    buf[: n_mine * DT.itemsize] = d.view(np.uint8)
                tm = torch.from_numpy(buf).to(tdev)
                parts = [torch.empty_like(tm) for _ in range(world)]
                dist.all_gather(parts, tm)
                alld = np.concatenate([parts[r][: counts[r] * DT.itemsize].cpu().numpy().view(DT) for r in range(world)]) if sum(counts) else np.zeros(0, DT)
                v = dupset.walk(path, alld)
                acc = (v & 8) != 0
                cum = np.cumsum(acc)
                want = nreads - total                                # accepted reads still wanted (> 0: the rounds stop when it reaches 0)
                end = len(v)                                         # records of the round the reference's sampler looks at
                if len(v) and cum[-1] >= want:
                    end = int(np.searchsorted(cum, want)) + 1        # ... up to and including the read that fills the sample
                    cut = True
                bad_at = np.nonzero((v[:end] & 64) != 0)[0]
                if len(bad_at):                                      # the reference raises here (every rank sees the same record)
                    f = int(v[bad_at[0]])
                    stats["error"] = ("KeyError: base outside ACGTN in reverse_complement" if f & 4 else
                                      "TypeError: record without qualities in a FASTQ run" if not f & 2 else "ValueError: empty quality string")
                    status = 3
                else:
                    ve = v[:end]
                    stats["too_short"] += int(np.count_nonzero(ve & 1)); stats["dups"] += int(np.count_nonzero(ve & 32))
                    stats["low_qual"] += int(np.count_nonzero((ve & 16) != 0)); stats["records"] += end; stats["bases"] += int(alld["len"][:end].sum())
                    lo_i = sum(counts[:rank]); hi_i = lo_i + n_mine
                    before = int(cum[lo_i - 1]) if lo_i > 0 else 0   # accepted reads of the round in front of this rank's slice
                    keep = int(np.count_nonzero(acc[lo_i:min(hi_i, end)]))
                    if keep > 0:
                        try:
                            on_batch(box["rd"].take(v[lo_i:hi_i], keep), total + before)
                        except BaseException as e:                   # noqa: BLE001 - agreed on below
                            err = e
                    total += int(cum[end - 1]) if end > 0 else 0
            if "rd" in box:
                box["rd"].close()
            flag = torch.tensor([1 if err is not None else 0], dtype=torch.int64, device=tdev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if int(flag.item()) and status == 0:
                status = 2
            if status or cut:
                break
    finally:
        if cur is not None:                                          # a round parsed ahead and not needed
            cur[0].join()
            if "rd" in cur[1]:
                cur[1]["rd"].close()
        dupset.close()
    bases = stats["bases"] if (status == 0 and not cut) else -1
    if status == 2:
        msgs = [None] * world
        dist.all_gather_object(msgs, "" if err is None else "rank %d: %s: %s" % (rank, type(err).__name__, err))
        stats["error"] = next((m for m in msgs if m), "a rank failed")
    return min(total, nreads), stats, bases, status


def run_pipeline_distributed(args, device=None):
    """run_pipeline() over all ranks of the initialised torch.distributed group (one process per GPU; backend "nccl" = RCCL
    on MI355X, or gloo), STREAMED.  Plain files without -d: a sampler on EVERY rank (stream_batches_sharded: byte windows of
    the files, the head-take agreed on by a prefix sum of the accepted counts).  Compressed inputs, -d, or a window that does not
    end on a record boundary: rank 0 runs the (sequential, deterministic) sampler on a thread of its own and deals batches of
    2 M accepted reads to whichever rank holds a credit (_Dealer; RCCL: pinned host -> GPU staging -> xGMI).  Either way
    every rank searches the batches it gets with global read ids while the next one arrives; the per-family integer
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
    threads = args.get("threads")
    if not threads and world > 1:
        # the ranks of one node share its CPUs (and, in a container, ONE cgroup quota: threads beyond it get every thread of every rank
        # throttled, the ones that drive the GPUs too - DESIGN.md 3): without an explicit -t each rank's sampler takes its share
        threads = max(2, _usable_cores() // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", world))))
    mc._cap_host_threads(threads)
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
        parts = []

        def on_batch(block, first):
            if isinstance(block, np.ndarray):                                  # host memory: rank 0's own batch
                eng.search(block, first_read_id=first)
            elif not nccl:                                                     # gloo: a host tensor
                eng.search(block.numpy().reshape(-1, L), first_read_id=first)
            else:                                                              # the batch is already in this GPU's HBM
                n = block.numel() // L
                eng.attach(block.data_ptr(), n)
                try:
                    eng.run_range(0, n, first_read_id=first)
                finally:
                    eng.attach(0, 0)
            parts.append(eng.best_hits())

        rd = None
        eng.lib.mc_set_keep_rows(eng.h, 0)
        sharded = None
        if world > 1 and (sharded_sampling_usable(args) or sharded_dups_usable(args)):   # a sampler on every rank (plain files; -d: the verdicts from exchanged descriptors)
            try:
                sharded = (stream_batches_sharded_dups if args.get("filter_dups") else stream_batches_sharded)(args, on_batch, device=device)
            finally:
                eng.lib.mc_set_keep_rows(eng.h, 1)
            if sharded[3] == 3:                                                # the reference raises at a record the sampler reached: run_pipeline prints it and returns None
                raise Exception(sharded[1]["error"])
            if sharded[3] == 2:                                                # a rank failed (sampler or search): all ranks raise the same error
                raise RuntimeError("sharded sampling failed - " + sharded[1].get("error", ""))
            if sharded[3] != 0:                                                # a window off a record boundary: the sampler on rank 0 decides
                sharded = None
                del parts[:]
                eng.lib.mc_set_keep_rows(eng.h, 0)
        try:
            if sharded is not None:
                n_total, trace, err, bases = sharded[0], None, None, sharded[2]
                run_pipeline_distributed.last_trace = None
                run_pipeline_distributed.last_stats = sharded[1]
            elif rank == 0:
                rd = _native.Reader(args["seqfiles"], L, args["nreads"], args["file_type"] == "fastq", args.get("quality_offset") or 0,
                                    args["min_quality"], args["mean_quality"], args["max_unknown"], args["filter_dups"])
            if sharded is None:
                n_total, trace, err = stream_batches(rd, L, on_batch, device=device)
                run_pipeline_distributed.last_trace = trace
                bases = -1
                if rank == 0 and err is None:
                    st = rd.stats()
                    bases = int(st["bases"]) if st.get("exhausted") else -1
        finally:
            eng.lib.mc_set_keep_rows(eng.h, 1)
            if rd is not None:
                rd.close()
        if err is not None:
            raise err
        head = [n_total, bases]
        if world > 1:
            dist.broadcast_object_list(head, src=0)
        n_total = head[0]
        if n_total == 0:
            sys.exit("\nError! No reads remaining after filtering!")
        args["sampled_reads"] = n_total
        if head[1] >= 0:
            mc._bases_cache[tuple(args["seqfiles"])] = head[1]
        run_pipeline_distributed.last_batches = len(parts)                       # (batches THIS rank searched: the tests look at the dealing)
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
