"""ctypes binding of libmcensus_hip.so (include/mcensus.h).  No CPU fallback: if the HIP library is
missing or no MI355X is visible, importing the engine fails loudly."""
import ctypes as C
import gzip
import json
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmcensus_hip.so")
DATA_DIR = os.path.join(_HERE, "data")


class McRow(C.Structure):
    _fields_ = [("query", C.c_int32), ("subject", C.c_int32), ("ident", C.c_double), ("alnlen", C.c_int32),
                ("mismatch", C.c_int32), ("gapopen", C.c_int32), ("qstart", C.c_int32), ("qend", C.c_int32),
                ("sstart", C.c_int32), ("send", C.c_int32), ("loge", C.c_double), ("bits", C.c_double),
                ("score", C.c_int32), ("nmatch", C.c_int32)]


class McBestHit(C.Structure):
    _fields_ = [("read", C.c_int32), ("family", C.c_int32), ("aln", C.c_int32), ("target_len", C.c_int32),
                ("bits", C.c_double)]


class McStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("reads", "seed_tasks", "gap_tasks", "hsps", "rows", "reads_with_rows", "classified", "bucket_lookups", "key_probes")] + \
               [(n, C.c_float) for n in ("ms_translate", "ms_seed", "ms_eval", "ms_gapped", "ms_sort", "ms_finish", "ms_total")] + \
               [("_pad", C.c_float)] + [(n, C.c_int64) for n in ("seed_exact_asks", "seed_wild_asks", "seed_pair_asks", "seed_probes", "range_splits")]


class McReaderStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("sampled", "too_short", "low_qual", "dups", "records", "bases", "exhausted", "ragged_end")]


ROW_DTYPE = np.dtype([("query", "<i4"), ("subject", "<i4"), ("ident", "<f8"), ("alnlen", "<i4"), ("mismatch", "<i4"),
                      ("gapopen", "<i4"), ("qstart", "<i4"), ("qend", "<i4"), ("sstart", "<i4"), ("send", "<i4"),
                      ("loge", "<f8"), ("bits", "<f8"), ("score", "<i4"), ("nmatch", "<i4")], align=True)
BEST_DTYPE = np.dtype([("read", "<i4"), ("family", "<i4"), ("aln", "<i4"), ("target_len", "<i4"), ("bits", "<f8")], align=True)
REC_DESC_DTYPE = np.dtype([("h1", "<u8"), ("h2", "<u8"), ("seq_off", "<u8"), ("len", "<u4"), ("flags", "u1"), ("pad", "u1", (3,))])   # mc_rec_desc

_lib = None


def _share_hip_runtime():
    """One HIP runtime per process.  A PyTorch-ROCm wheel brings its own copy of libamdhip64 (file name libamdhip64.so, SONAME
    libamdhip64.so.7); this library needs libamdhip64.so.7.  With torch imported first the loader hands this library torch's copy (the
    SONAME matches) and all is well; the other way round torch asks for "libamdhip64.so", which matches nothing loaded, gets its own
    copy beside the system's, and the second runtime fails to initialise the GPU.  So: where a torch with a HIP
    runtime of its own is installed and not yet imported, that copy is loaded first and both use it - in whatever order the application
    then imports them.  MCENSUS_HIP_RUNTIME=system in the environment: leave it to the loader."""
    if os.environ.get("MCENSUS_HIP_RUNTIME", "") == "system" or "torch" in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so") if spec is not None and spec.origin else None
        if cand and os.path.isfile(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:                                   # (no torch, or one without a runtime of its own: the system's it is)
        pass


def load_library():
    """Load libmcensus_hip.so and declare the prototypes of include/mcensus.h."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("MCENSUS_LIB", LIB_PATH)      # development: an alternative build of the same library
    if not os.path.isfile(path):
        raise RuntimeError("HIP extension %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'`" % path)
    _share_hip_runtime()
    lib = C.CDLL(path)
    lib.mc_last_error.restype = C.c_char_p
    lib.mc_device_count.restype = C.c_int
    lib.mc_open.restype = C.c_void_p
    lib.mc_open.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.c_int32]
    lib.mc_close.argtypes = [C.c_void_p]
    lib.mc_set_index_cache.argtypes = [C.c_char_p]
    lib.mc_index_cache_check.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_int32, C.c_char_p]
    lib.mc_open_rapdb.restype = C.c_void_p
    lib.mc_open_rapdb.argtypes = [C.c_char_p, C.c_int32]
    lib.mc_marker_count.restype = C.c_int32
    lib.mc_marker_count.argtypes = [C.c_void_p]
    lib.mc_marker_name.restype = C.c_char_p
    lib.mc_marker_name.argtypes = [C.c_void_p, C.c_int32]
    lib.mc_set_families.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int32]
    lib.mc_rapdb_verify.argtypes = [C.c_char_p, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_int32]
    lib.mc_rapdb_write.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_int32, C.c_char_p]
    lib.mc_index_view.argtypes = [C.c_void_p] + [C.POINTER(C.c_void_p)] * 5 + [C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_uint32), C.POINTER(C.c_double)]
    lib.mc_set_run.argtypes = [C.c_void_p, C.c_int32, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.mc_search.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64]
    lib.mc_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    lib.mc_run.argtypes = [C.c_void_p, C.c_int64]
    lib.mc_attach.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    lib.mc_run_range.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64]
    lib.mc_set_counting.argtypes = [C.c_void_p, C.c_int]
    lib.mc_debug_stage.restype = C.c_int64
    lib.mc_debug_stage.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_int32)]
    lib.mc_range_begin.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64]
    lib.mc_range_end.argtypes = [C.c_void_p]
    lib.mc_ranges_in_flight.argtypes = [C.c_void_p]
    lib.mc_result_rows.restype = C.c_int64
    lib.mc_result_rows.argtypes = [C.c_void_p, C.POINTER(C.POINTER(McRow))]
    lib.mc_result_best_hits.restype = C.c_int64
    lib.mc_result_best_hits.argtypes = [C.c_void_p, C.POINTER(C.POINTER(McBestHit))]
    lib.mc_result_stats.argtypes = [C.c_void_p, C.POINTER(McStats)]
    lib.mc_write_m8.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    lib.mc_write_m8_named.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_char_p), C.c_int64, C.c_int64]
    lib.mc_reader_last_error.restype = C.c_char_p
    lib.mc_set_host_threads.restype = None
    lib.mc_set_host_threads.argtypes = [C.c_int32]
    lib.mc_reader_open.restype = C.c_void_p
    lib.mc_reader_open.argtypes = [C.POINTER(C.c_char_p), C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_int32, C.c_char_p]
    lib.mc_reader_open_range.restype = C.c_void_p
    lib.mc_reader_open_range.argtypes = [C.c_char_p, C.c_int64, C.c_int64, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double]
    lib.mc_reader_run.restype = C.c_int64
    lib.mc_reader_run.argtypes = [C.c_void_p]
    lib.mc_reader_reads.restype = C.POINTER(C.c_uint8)
    lib.mc_reader_reads.argtypes = [C.c_void_p]
    lib.mc_reader_get_stats.argtypes = [C.c_void_p, C.POINTER(McReaderStats)]
    lib.mc_reader_open_bz2_part.restype = C.c_void_p
    lib.mc_reader_open_bz2_part.argtypes = [C.c_char_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double]
    lib.mc_bz2_blocks.restype = C.c_int64
    lib.mc_bz2_blocks.argtypes = [C.c_char_p]
    lib.mc_gz_chunks.restype = C.c_int64
    lib.mc_gz_chunks.argtypes = [C.c_char_p, C.c_int64]
    lib.mc_reader_open_gz_part.restype = C.c_void_p
    lib.mc_reader_open_gz_part.argtypes = [C.c_char_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double]
    lib.mc_reader_gz_provide.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    lib.mc_reader_gz_end_state.restype = C.c_int64
    lib.mc_reader_gz_end_state.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    lib.mc_reader_gz_finish.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
    lib.mc_reader_describe.restype = C.c_int64
    lib.mc_reader_describe.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    lib.mc_dupset_open.restype = C.c_void_p
    lib.mc_dupset_close.argtypes = [C.c_void_p]
    lib.mc_dupset_walk.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64, C.c_void_p]
    lib.mc_reader_take.restype = C.c_int64
    lib.mc_reader_take.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
    lib.mc_reader_times.restype = C.c_int32
    lib.mc_reader_times.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int32]
    lib.mc_reader_close.argtypes = [C.c_void_p]
    lib.mc_reader_trim.restype = None
    lib.mc_reader_trim.argtypes = [C.c_int64]
    lib.mc_count_bases.restype = C.c_int64
    lib.mc_count_bases.argtypes = [C.POINTER(C.c_char_p), C.c_int32]
    lib.mc_quality_offset.restype = C.c_int32
    lib.mc_quality_offset.argtypes = [C.c_char_p]
    lib.mc_reader_start.argtypes = [C.c_void_p]
    lib.mc_reader_fetch.restype = C.c_int64
    lib.mc_reader_fetch.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
    lib.mc_reader_join.restype = C.c_int64
    lib.mc_reader_join.argtypes = [C.c_void_p]
    lib.mc_reader_read_len.restype = C.c_int32
    lib.mc_reader_read_len.argtypes = [C.c_void_p]
    lib.mc_reader_nreads.restype = C.c_int64
    lib.mc_reader_nreads.argtypes = [C.c_void_p]
    lib.mc_search_files.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    lib.mc_search_files_multi.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.c_void_p, C.c_int64]
    lib.mc_set_keep_rows.argtypes = [C.c_void_p, C.c_int]
    lib.mc_set_best_hits_only.argtypes = [C.c_void_p, C.c_int]
    lib.mc_grid_classify.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_double), C.c_int32,
                                     C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_double)]
    _lib = lib
    return lib


EXPORTED_SYMBOLS = ["mc_last_error", "mc_device_count", "mc_open", "mc_close", "mc_set_index_cache", "mc_index_cache_check", "mc_open_rapdb", "mc_marker_count", "mc_marker_name", "mc_set_families", "mc_rapdb_verify", "mc_rapdb_write", "mc_index_view", "mc_set_run", "mc_search",
                    "mc_upload", "mc_attach", "mc_run", "mc_run_range", "mc_set_counting", "mc_debug_stage", "mc_range_begin", "mc_range_end", "mc_ranges_in_flight", "mc_result_rows", "mc_result_best_hits", "mc_result_stats", "mc_write_m8", "mc_write_m8_named",
                    "mc_reader_last_error", "mc_set_host_threads", "mc_reader_open", "mc_reader_open_range", "mc_reader_open_bz2_part", "mc_bz2_blocks", "mc_gz_chunks", "mc_reader_open_gz_part", "mc_reader_gz_provide", "mc_reader_gz_end_state", "mc_reader_gz_finish", "mc_reader_run", "mc_reader_reads", "mc_reader_get_stats", "mc_reader_times", "mc_reader_describe", "mc_dupset_open", "mc_dupset_close", "mc_dupset_walk", "mc_reader_take", "mc_reader_close", "mc_reader_trim", "mc_count_bases", "mc_quality_offset",
                    "mc_reader_start", "mc_reader_fetch", "mc_reader_join", "mc_reader_read_len", "mc_reader_nreads", "mc_search_files", "mc_search_files_multi", "mc_set_keep_rows", "mc_set_best_hits_only", "mc_grid_classify"]


class DupSet:
    """The sampler's set of accepted sequences for -d, walked over record descriptors (mc_dupset_*): verdicts in file order."""

    def __init__(self):
        self.lib = load_library()
        self.s = self.lib.mc_dupset_open()
        if not self.s:
            raise RuntimeError(self.lib.mc_reader_last_error().decode())

    def walk(self, path, descs):
        import numpy as np
        d = np.ascontiguousarray(descs)
        v = np.empty(len(d), np.uint8)
        if self.lib.mc_dupset_walk(self.s, path.encode(), d.ctypes.data, len(d), v.ctypes.data) != 0:
            raise RuntimeError(self.lib.mc_reader_last_error().decode())
        return v

    def close(self):
        if getattr(self, "s", None):
            self.lib.mc_dupset_close(self.s)
            self.s = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ReferenceError_(Exception):
    """The reference's Python would have raised inside run_pipeline (which prints the error and returns None)."""


class Reader:
    """The native sampler (mc_reader_*): process_seqfile's rules on plain / .gz / .bz2 FASTA / FASTQ files.  Needs no GPU."""

    def __init__(self, paths, read_len, nreads, fastq, quality_offset, min_quality, mean_quality, max_unknown, filter_dups, fasta_out=None):
        lib = load_library()
        if nreads is None:                   # the reference's "no cap" (its `read_id == args['nreads']` is never true for None)
            nreads = (1 << 63) - 1
        arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
        self.lib, self.read_len = lib, read_len
        self.r = lib.mc_reader_open(arr, len(paths), read_len, nreads, 1 if fastq else 0, int(quality_offset), float(min_quality), float(mean_quality),
                                    float(max_unknown), 1 if filter_dups else 0, fasta_out.encode() if fasta_out else None)
        if not self.r:
            raise RuntimeError(lib.mc_reader_last_error().decode())

    @classmethod
    def on_range(cls, path, byte_lo, byte_hi, read_len, nreads, fastq, quality_offset, min_quality, mean_quality, max_unknown):
        """The sampler on the records that start in [byte_lo, byte_hi) of one plain file (mc_reader_open_range)."""
        lib = load_library()
        self = cls.__new__(cls)
        self.lib, self.read_len = lib, read_len
        if nreads is None:
            nreads = (1 << 63) - 1
        self.r = lib.mc_reader_open_range(path.encode(), int(byte_lo), int(byte_hi), read_len, nreads, 1 if fastq else 0, int(quality_offset), float(min_quality),
                                          float(mean_quality), float(max_unknown))
        if not self.r:
            raise RuntimeError(lib.mc_reader_last_error().decode())
        return self

    @classmethod
    def on_bz2_part(cls, path, block_lo, block_hi, kind, read_len, nreads, fastq, quality_offset, min_quality, mean_quality, max_unknown):
        """The sampler on the records that start in the text of blocks [block_lo, block_hi) of one .bz2 file (mc_reader_open_bz2_part)."""
        lib = load_library()
        self = cls.__new__(cls)
        self.lib, self.read_len = lib, read_len
        if nreads is None:
            nreads = (1 << 63) - 1
        self.r = lib.mc_reader_open_bz2_part(path.encode(), int(block_lo), int(block_hi), ord(kind), read_len, nreads, 1 if fastq else 0, int(quality_offset),
                                             float(min_quality), float(mean_quality), float(max_unknown))
        if not self.r:
            raise RuntimeError(lib.mc_reader_last_error().decode())
        return self

    @classmethod
    def on_gz_part(cls, path, chunk_lo, chunk_hi, chunk_bytes, kind, read_len, nreads, fastq, quality_offset, min_quality, mean_quality, max_unknown):
        """The sampler on the records that start in the text of chunks [chunk_lo, chunk_hi) of one .gz file (mc_reader_open_gz_part): start() it,
        gz_provide() the state of the slice in front (not for chunk 0), gz_end_state() for the next slice's owner, join(), gz_finish()."""
        lib = load_library()
        self = cls.__new__(cls)
        self.lib, self.read_len = lib, read_len
        if nreads is None:
            nreads = (1 << 63) - 1
        self.r = lib.mc_reader_open_gz_part(path.encode(), int(chunk_lo), int(chunk_hi), int(chunk_bytes), ord(kind), read_len, nreads, 1 if fastq else 0, int(quality_offset),
                                            float(min_quality), float(mean_quality), float(max_unknown))
        if not self.r:
            raise RuntimeError(lib.mc_reader_last_error().decode())
        return self

    def start(self):
        if self.lib.mc_reader_start(self.r) != 0:
            raise RuntimeError(self.lib.mc_reader_last_error().decode())

    def join(self):
        return self.check(self.lib.mc_reader_join(self.r))

    def gz_provide(self, state):
        self.lib.mc_reader_gz_provide(self.r, state if state else None, len(state) if state else 0)

    def gz_end_state(self):
        """bytes for the owner of the next slice, or None when this slice failed"""
        buf = C.create_string_buffer(32768 + 64)
        n = self.lib.mc_reader_gz_end_state(self.r, buf, len(buf))
        return None if n < 0 else buf.raw[:n]

    def gz_finish(self, crc_in):
        """crc_in / result: 12 bytes (CRC | length of the open member's bytes so far); raises ReferenceError_ on a CRC mismatch"""
        out = C.create_string_buffer(12)
        self.check(self.lib.mc_reader_gz_finish(self.r, crc_in, out))
        return out.raw

    def close(self):
        if getattr(self, "r", None):
            self.lib.mc_reader_close(self.r)
            self.r = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, n):
        if n == -3:
            raise ReferenceError_(self.lib.mc_reader_last_error().decode())
        if n < 0:
            raise RuntimeError(self.lib.mc_reader_last_error().decode())
        return n

    def run(self):
        return self.check(self.lib.mc_reader_run(self.r))

    def stats(self):
        st = McReaderStats()
        self.lib.mc_reader_get_stats(self.r, C.byref(st))
        return {k: getattr(st, k) for k, _ in McReaderStats._fields_}

    def describe(self):
        """The records of the window of a reader opened with on_range, one 32-byte descriptor each (mc_reader_describe): a numpy
        structured array (a copy: it travels to the other ranks)."""
        import numpy as np
        p = C.c_void_p()
        n = self.check(self.lib.mc_reader_describe(self.r, C.byref(p)))
        if n == 0:
            return np.zeros(0, REC_DESC_DTYPE)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n * REC_DESC_DTYPE.itemsize,)).view(REC_DESC_DTYPE).copy()

    def take(self, verdicts, max_take):
        """(k, read_len) uint8: the accepted reads of the described window (verdict bit 8), at most max_take (mc_reader_take)."""
        import numpy as np
        v = np.ascontiguousarray(verdicts, dtype=np.uint8)
        k = int(min(max_take, int(np.count_nonzero(v & 8))))
        out = np.empty((k, self.read_len), np.uint8)
        got = self.check(self.lib.mc_reader_take(self.r, v.ctypes.data, len(v), k, out.ctypes.data))
        return out[:got]

    def times(self):
        """Seconds of the last run by phase (mc_reader_times)."""
        v = (C.c_double * 7)()
        k = self.lib.mc_reader_times(self.r, v, 7)
        names = ("run", "input_wait", "guesses", "parse", "stitch", "verdicts_places_copies", "dup_class_walkers")
        return {names[i]: round(v[i], 4) for i in range(max(k, 0))}

    def reads(self, n):
        """(n, read_len) uint8 view of the sampled reads; it keeps the reader alive."""
        if not n:
            return np.zeros((0, self.read_len), np.uint8)
        buf = (C.c_uint8 * (n * self.read_len)).from_address(C.addressof(self.lib.mc_reader_reads(self.r).contents))
        buf._owner = self
        a = np.frombuffer(buf, dtype=np.uint8).reshape(n, self.read_len)
        return a


def sample_reads(paths, read_len, nreads, fastq, quality_offset, min_quality, mean_quality, max_unknown, filter_dups, fasta_out=None):
    """Native process_seqfile: returns (reads uint8 (n, read_len), stats dict).  Needs no GPU."""
    rd = Reader(paths, read_len, nreads, fastq, quality_offset, min_quality, mean_quality, max_unknown, filter_dups, fasta_out)
    n = rd.run()
    return rd.reads(n), rd.stats()


def count_bases(paths):
    lib = load_library()
    arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
    n = lib.mc_count_bases(arr, len(paths))
    if n == -3:
        raise ReferenceError_(lib.mc_reader_last_error().decode())
    if n < 0:
        raise RuntimeError(lib.mc_reader_last_error().decode())
    return n


def gz_chunks(path, chunk_bytes=1 << 20):
    """Chunks the parallel gzip reader cuts a .gz file into (mc_gz_chunks), or -1."""
    return int(load_library().mc_gz_chunks(path.encode(), int(chunk_bytes)))


def bz2_blocks(path):
    """Blocks of a .bz2 file all of whose streams check out (mc_bz2_blocks), or -1."""
    return int(load_library().mc_bz2_blocks(path.encode()))


def quality_offset(path):
    """auto_detect_quality_offset by the native parser: 32 or 64; None when the Python path has to decide (a record without
    qualities, an unreadable file: it then fails the way the reference does)."""
    v = load_library().mc_quality_offset(path.encode())
    return v if v in (32, 64) else None


def rapdb_verify(rapdb_path, names, seqs):
    """0 when the prerapsearch database holds exactly the index mc_open() builds from (names, seqs).  No GPU involved."""
    lib = load_library()
    n = len(names)
    return lib.mc_rapdb_verify(rapdb_path.encode(), (C.c_char_p * n)(*[s.encode() for s in names]), (C.c_char_p * n)(*[s.encode() for s in seqs]), n)


def rapdb_write(names, seqs, path):
    """Writes <path> and <path>.info as `prerapsearch -d markers.faa -n <path>` would.  No GPU involved."""
    lib = load_library()
    n = len(names)
    if lib.mc_rapdb_write((C.c_char_p * n)(*[s.encode() for s in names]), (C.c_char_p * n)(*[s.encode() for s in seqs]), n, path.encode()) != 0:
        raise RuntimeError(lib.mc_last_error().decode())


def load_markers(path=None):
    """(names, seqs) of the canonical marker FASTA shipped as package data."""
    path = path or os.path.join(DATA_DIR, "markers.faa.gz")
    names, seqs = [], []
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rt") as f:
        for line in f:
            if line.startswith(">"):
                names.append(line[1:].split()[0])
                seqs.append([])
            else:
                seqs[-1].append(line.strip())
    return names, ["".join(s) for s in seqs]


def load_model(path=None):
    with open(path or os.path.join(DATA_DIR, "model.json")) as f:
        return json.load(f)


ALN_STAT = {"hits": 0, "cov": 1, "aln": 2}


def user_cache_dir():
    """~/.cache/microbecensus_amd (XDG_CACHE_HOME honoured), created with mode 0700; None when it cannot be made or is not this
    user's alone (owned by somebody else, or group / world writable) - nothing is cached then."""
    root = os.path.join(os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache"), "microbecensus_amd")
    return _private_dir(root)


def _private_dir(root):
    """root, made with mode 0700 if it is not there; None when it cannot be made or is not this user's alone."""
    try:
        os.makedirs(root, mode=0o700, exist_ok=True)
        st = os.stat(root)
        if st.st_uid != os.getuid() or (st.st_mode & 0o022):
            return None
    except OSError:
        return None
    return root


_index_cache_set = False


def use_index_cache():
    """Points mc_open() at the per-user cache of built indexes (once per process; MC_INDEX_CACHE=0 in the environment: no cache,
    MC_INDEX_CACHE=<dir>: that directory - like the default one only if it is this user's alone: a cached index is
    read back into device arrays the kernels index, and although mc_open compares it with its inputs and checks its offsets
    (mc_index_matches_input), a directory others can write is not a place to take it from)."""
    global _index_cache_set
    if _index_cache_set:
        return
    _index_cache_set = True
    v = os.environ.get("MC_INDEX_CACHE")
    if v == "0":
        return
    d = _private_dir(v) if v else user_cache_dir()      # (a directory named in the environment passes the same test: this user's, not group / world writable)
    if d:
        load_library().mc_set_index_cache(d.encode())


class Engine:
    """One MI355X: marker index resident in HBM + the search/classify pipeline."""

    def __init__(self, device=0, names=None, seqs=None, marker_family=None, nfam=None):
        lib = load_library()
        use_index_cache()
        if names is None:
            names, seqs = load_markers()
            model = load_model()
            marker_family, nfam = model["marker_family"], len(model["families"])
        self.lib, self.names, self.nfam = lib, names, nfam
        n = len(names)
        an = (C.c_char_p * n)(*[s.encode() for s in names])
        asq = (C.c_char_p * n)(*[s.encode() for s in seqs])
        fam = (C.c_int32 * n)(*marker_family)
        self.h = lib.mc_open(an, asq, n, fam, nfam, device)
        if not self.h:
            raise RuntimeError("mc_open failed: %s" % lib.mc_last_error().decode())
        self.read_len = None

    @classmethod
    def from_rapdb(cls, rapdb_path, device=0, family_of=None, families=None):
        """Engine on a database written by prerapsearch (e.g. the reference's data/rapdb_2.15).  family_of: marker name ->
        family name (gene_fam.map); default: the packaged model's."""
        lib = load_library()
        self = cls.__new__(cls)
        self.lib = lib
        self.h = lib.mc_open_rapdb(rapdb_path.encode(), device)
        if not self.h:
            raise RuntimeError("mc_open_rapdb failed: %s" % lib.mc_last_error().decode())
        n = lib.mc_marker_count(self.h)
        self.names = [lib.mc_marker_name(self.h, i).decode() for i in range(n)]
        if family_of is None:
            model = load_model()
            pk_names, _ = load_markers()
            families = model["families"]
            fam_idx = dict(zip(pk_names, model["marker_family"]))
            fam = [fam_idx.get(nm, 0) for nm in self.names]       # (a database of other sequences: one family; searching needs none)
        else:
            fam = [families.index(family_of[nm]) for nm in self.names]
        self.nfam = len(families)
        self._check(lib.mc_set_families(self.h, (C.c_int32 * n)(*fam), self.nfam), "mc_set_families")
        self.read_len = None
        return self

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed (%d): %s" % (what, rc, self.lib.mc_last_error().decode()))

    def close(self):
        if getattr(self, "h", None):
            self.lib.mc_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_run(self, read_len, pars=None, families=None, loge_thr=1.0):
        """pars: {family: [min_cov, max_aaid, min_score, aln_stat]} as find_opt_pars returns for this length."""
        nf = self.nfam
        cov = (C.c_double * nf)(*([0.0] * nf)); score = (C.c_double * nf)(*([0.0] * nf))
        aaid = (C.c_int32 * nf)(*([100] * nf)); stat = (C.c_int32 * nf)(*([0] * nf))
        if pars is not None:
            for i, fam in enumerate(families):
                p = pars[fam]
                cov[i], aaid[i], score[i], stat[i] = float(p[0]), int(round(float(p[1]))), float(p[2]), ALN_STAT[p[3]]
                assert float(p[1]) == aaid[i]
        self._check(self.lib.mc_set_run(self.h, read_len, loge_thr, cov, score, aaid, stat), "mc_set_run")
        self.read_len = read_len

    def search(self, reads, first_read_id=0):
        """reads: uint8 array (n, read_len) of bases.  Returns (rows, best_hits) as numpy structured arrays."""
        reads = np.ascontiguousarray(reads, dtype=np.uint8)
        assert reads.ndim == 2 and reads.shape[1] == self.read_len
        self._check(self.lib.mc_search(self.h, reads.ctypes.data_as(C.c_void_p), reads.shape[0], first_read_id), "mc_search")
        return self.results()

    def set_best_hits_only(self, on):
        """Only the reads that can be classified are ranked (mc_set_best_hits_only): same best hits, no rows."""
        self._check(self.lib.mc_set_best_hits_only(self.h, 1 if on else 0), "mc_set_best_hits_only")

    def search_files(self, reader, first_read_id=0, keep_rows=True, best_only=False):
        """process_seqfile + search_seqs + classify_reads in one call (mc_search_files): the reader samples beside the search."""
        self._check(self.lib.mc_set_keep_rows(self.h, 1 if keep_rows else 0), "mc_set_keep_rows")
        self.set_best_hits_only(best_only)
        try:
            rc = self.lib.mc_search_files(self.h, reader.r, first_read_id)
            if rc == -3:
                raise ReferenceError_(self.lib.mc_last_error().decode())
            self._check(rc, "mc_search_files")
        finally:
            self.lib.mc_set_keep_rows(self.h, 1)
            self.lib.mc_set_best_hits_only(self.h, 0)
        return self.results()

    def upload(self, reads):
        reads = np.ascontiguousarray(reads, dtype=np.uint8)
        assert reads.ndim == 2 and reads.shape[1] == self.read_len
        self._check(self.lib.mc_upload(self.h, reads.ctypes.data_as(C.c_void_p), reads.shape[0]), "mc_upload")

    def run(self, first_read_id=0):
        self._check(self.lib.mc_run(self.h, first_read_id), "mc_run")

    def attach(self, device_ptr, nreads):
        """Adopt caller-owned device memory (nreads x read_len bytes) as the resident read set."""
        self._check(self.lib.mc_attach(self.h, C.c_void_p(device_ptr), nreads), "mc_attach")

    def set_counting(self, on):
        """Turns the seed kernel's algorithmic-traffic counters (stats bucket_lookups / key_probes) on or off."""
        self._check(self.lib.mc_set_counting(self.h, 1 if on else 0), "mc_set_counting")

    def range_begin(self, first, count, first_read_id=0):
        """Enqueues the front of the range (translation, seeds, seed evaluation) and returns at once.  range_end(), range_begin(next),
        then the results of the range that ended: the device runs the next front while the host looks at them.  One range at a time."""
        self._check(self.lib.mc_range_begin(self.h, first, count, first_read_id), "mc_range_begin")

    def range_end(self):
        """Completes the range in flight: rows() / best_hits() / stats() are its results."""
        self._check(self.lib.mc_range_end(self.h), "mc_range_end")

    def ranges_in_flight(self):
        return self.lib.mc_ranges_in_flight(self.h)

    def run_range(self, first, count, first_read_id=0):
        self._check(self.lib.mc_run_range(self.h, first, count, first_read_id), "mc_run_range")

    def debug_stage(self, what):
        """Test aid: what a stage of the last run_range() left on the device (0 frames, 1 seed hits, 2 gap tasks, 3 HSP pool) as a
        (records, record_bytes) uint8 array."""
        rec = C.c_int32(0)
        n = self.lib.mc_debug_stage(self.h, what, None, 0, C.byref(rec))
        if n < 0:
            raise RuntimeError("mc_debug_stage failed: %s" % self.lib.mc_last_error().decode())
        buf = np.zeros(n, np.uint8)
        if n and self.lib.mc_debug_stage(self.h, what, buf.ctypes.data_as(C.c_void_p), n, C.byref(rec)) != n:
            raise RuntimeError("mc_debug_stage failed: %s" % self.lib.mc_last_error().decode())
        return buf.reshape(-1, rec.value) if rec.value else buf

    def rows(self, copy=True):
        """m8 rows of the last run (structured array).  copy=False returns a view of the handle's buffer, valid until the next call."""
        pr = C.POINTER(McRow)()
        n = self.lib.mc_result_rows(self.h, C.byref(pr))
        if not n:
            return np.zeros(0, ROW_DTYPE)
        a = np.ctypeslib.as_array(C.cast(pr, C.POINTER(C.c_uint8)), shape=(n * C.sizeof(McRow),)).view(ROW_DTYPE)
        return a.copy() if copy else a

    def best_hits(self, copy=True):
        """best hit per classified read of the last run, ascending read id."""
        pb = C.POINTER(McBestHit)()
        m = self.lib.mc_result_best_hits(self.h, C.byref(pb))
        if not m:
            return np.zeros(0, BEST_DTYPE)
        a = np.ctypeslib.as_array(C.cast(pb, C.POINTER(C.c_uint8)), shape=(m * C.sizeof(McBestHit),)).view(BEST_DTYPE)
        return a.copy() if copy else a

    def results(self):
        return self.rows(), self.best_hits()

    def stats(self):
        s = McStats()
        self._check(self.lib.mc_result_stats(self.h, C.byref(s)), "mc_result_stats")
        return {k: getattr(s, k) for k, _ in McStats._fields_ if not k.startswith("_")}

    def write_m8(self, path, append=False):
        self._check(self.lib.mc_write_m8(self.h, path.encode(), 1 if append else 0), "mc_write_m8")

    def write_m8_named(self, path, names, append=False, first_read_id=0):
        arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
        self._check(self.lib.mc_write_m8_named(self.h, path.encode(), 1 if append else 0, arr, len(names), first_read_id), "mc_write_m8_named")

    def grid_classify(self, aln_covs, max_pids, min_scores):
        """The training grid (training/training.py:311-334) over the rows of the last search: (hits, aln, cov) arrays of shape
        (len(aln_covs), len(max_pids), len(min_scores), nfam)."""
        nc, npid, ns = len(aln_covs), len(max_pids), len(min_scores)
        shape = (nc, npid, ns, self.nfam)
        hits = np.zeros(shape, np.int64); aln = np.zeros(shape, np.int64); cov = np.zeros(shape, np.float64)
        self._check(self.lib.mc_grid_classify(self.h, (C.c_double * nc)(*aln_covs), nc, (C.c_int32 * npid)(*[int(p) for p in max_pids]), npid, (C.c_double * ns)(*min_scores), ns,
                                              hits.ctypes.data_as(C.POINTER(C.c_int64)), aln.ctypes.data_as(C.POINTER(C.c_int64)), cov.ctypes.data_as(C.POINTER(C.c_double))),
                    "mc_grid_classify")
        return hits, aln, cov

    def index_view(self):
        p = [C.c_void_p() for _ in range(5)]
        nres, npost, thr = C.c_int64(), C.c_int64(), C.c_uint32()
        lp = (C.c_double * 10)()
        self._check(self.lib.mc_index_view(self.h, *[C.byref(x) for x in p], C.byref(nres), C.byref(npost), C.byref(thr), lp), "mc_index_view")
        nseq = len(self.names)

        def arr(ptr, dtype, n):
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(n * np.dtype(dtype).itemsize,)).view(dtype).copy()
        return {"res": arr(p[0], "u1", nres.value), "off": arr(p[1], "<u4", nseq + 1), "bstart": arr(p[2], "<u4", 1000001),
                "post": arr(p[3], "<u4", npost.value), "keys": arr(p[4], "<u2", npost.value), "thr": thr.value, "letter_p": list(lp)}


def search_files_multi(engines, reader, first_read_id=0, keep_rows=False, best_only=False):
    """mc_search_files_multi: one sampler, its batches dealt to several engines (GPUs) of this process.  Returns the best hits of
    all engines in ascending read order."""
    lib = load_library()
    arr = (C.c_void_p * len(engines))(*[e.h for e in engines])
    for e in engines:
        lib.mc_set_keep_rows(e.h, 1 if keep_rows else 0)
        lib.mc_set_best_hits_only(e.h, 1 if best_only else 0)
    try:
        rc = lib.mc_search_files_multi(arr, len(engines), reader.r, first_read_id)
        if rc == -3:
            raise ReferenceError_(lib.mc_last_error().decode())
        if rc != 0:
            raise RuntimeError("mc_search_files_multi failed (%d): %s" % (rc, lib.mc_last_error().decode()))
    finally:
        for e in engines:
            lib.mc_set_keep_rows(e.h, 1)
            lib.mc_set_best_hits_only(e.h, 0)
    best = np.concatenate([e.best_hits() for e in engines])
    return best[np.argsort(best["read"], kind="stable")]
