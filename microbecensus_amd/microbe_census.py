"""Host-side mirror of the reference's module API (microbe_census/microbe_census.py of
snayfach/MicrobeCensus) with the hot path on an MI355X.

Same public names, argument dicts, side effects, messages and return shapes as the reference, so
`from microbecensus_amd import microbe_census` is a drop-in for `from microbe_census import microbe_census`:

    run_pipeline(args) -> (est_ags, args)            reference :586-631
    process_seqfile(args, paths)                     reference :328-367   (sampler / QC / trimming)
    search_seqs(args, paths)                         reference :369-389   -> HIP pipeline instead of rapsearch
    classify_reads(args, paths) -> best_hits         reference :432-460   -> device classification
    aggregate_hits(args, paths, best_hits)           reference :462-472
    estimate_average_genome_size(args, paths, agg)   reference :474-512
    count_bases(args), report_results(...)           reference :573-584, :514-529

What differs from the reference, on purpose:
  * search_seqs() does not fork RAPsearch2; it packs the trimmed reads and calls libmcensus_hip.so
    (include/mcensus.h).  It still leaves `<tempfile>.m8` behind in RAPsearch2's format, so anything that
    consumed that file keeps working.  args['rapsearch'] (the reference's -r hook) runs that executable instead,
    exactly as the reference does, on a database the library writes in RAPsearch2's format: the bundled CPU binary
    for an A/B run, or scripts/rapsearch_mi355x (the GPU engine behind RAPsearch2's command line).
  * classify_reads() takes the per-read best hits the device computed; if `<tempfile>.m8` was not produced
    by this process it falls back to parsing that file exactly as the reference does.
  * `.bz2` inputs are read (the reference returns a bytes stream under Python 3, which makes its own
    file-type detection exit).
  * run_pipeline() samples, searches and classifies in one native call (mc_search_files) and writes neither the
    temp FASTA nor the m8 text - the reference deletes both before it returns; args['keep_tmp'] = True runs stage by
    stage and leaves them (until clean_up) like the reference.
  * optional args['device'] selects ONE GPU; args['devices'] (a list of indices, or 'all') several: the sampler's batches of 2 M reads
    are dealt to them inside one process (mc_search_files_multi).  With neither, every visible GPU is used - as far as the run has
    batches for them (-n up to 2 M reads: one GPU).  Results do not depend on the number of GPUs.
  * args['threads'] (-t), when given, caps the worker threads of the native sampler (the reference forwards it to rapsearch -z).
"""
import bz2
import gzip
import io
import os
import platform
import sys
from tempfile import mkstemp

import numpy as np
from numpy import mean, median

__version__ = "1.1.0"

VALID_READ_LENGTHS = [50, 60, 70, 80, 90, 100, 110, 120, 130, 140, 150, 175, 200, 225, 250, 300, 350, 400, 450, 500]

_engines = {}       # device -> Engine
_run_cache = {}          # tempfile -> dict(reads=ndarray | None, best=ndarray | None, families=[...])
_bases_cache = {}        # seqfiles -> total bases, when the sampler happened to read every record


# ----------------------------------------------------------------------------------------------
# small helpers with the reference's semantics
# ----------------------------------------------------------------------------------------------
def mad(x, const=1.48):
    """const * median absolute deviation (reference :43-45)."""
    m = median(x)
    return const * median([abs(v - m) for v in x])


def open_file(inpath):
    """Text stream over a plain, .gz or .bz2 file (reference :47-59)."""
    ext = inpath.split(".")[-1]
    if ext == "gz":
        return io.TextIOWrapper(gzip.open(inpath))
    if ext == "bz2":
        return io.TextIOWrapper(bz2.BZ2File(inpath))
    return open(inpath)


def _model():
    from . import _native
    if "model" not in _run_cache:
        _run_cache["model"] = _native.load_model()
    return _run_cache["model"]


def find_opt_pars(path_optpars, read_length):
    """{family: {'min_cov','max_aaid','min_score','aln_stat'}} for one read length (reference :61-72)."""
    pars = _model()["pars"].get(str(read_length), {})
    return {fam: {"min_cov": p[0], "max_aaid": p[1], "min_score": p[2], "aln_stat": p[3]} for fam, p in pars.items()}


def read_list(file, header, dtype):
    """One value per line (reference :90-99)."""
    conv = float if dtype == "float" else int if dtype == "int" else (lambda v: v)
    with open_file(file) as f_in:
        if header is True:
            next(f_in)
        return [conv(line.rstrip()) for line in f_in]


def check_os():
    if platform.system() not in ["Linux", "Darwin"]:
        sys.exit("Operating system '%s' not supported" % platform.system())


def get_relative_paths(args):
    """Data locations + a fresh temp file (reference :106-123); the maps live in package data."""
    pkg_dir = os.path.dirname(os.path.abspath(__file__))
    paths = {"db": os.path.join(pkg_dir, "data", "markers.faa.gz"), "model": os.path.join(pkg_dir, "data", "model.json"),
             "tempfile": mkstemp()[1]}
    if args.get("rapsearch"):                     # the reference's -r hook (:110-111): an external RAPsearch2-compatible executable
        paths["rapsearch"] = args["rapsearch"]
    return paths


def check_rapsearch(rapsearch):
    """The executable must answer `-h` like RAPsearch2 v2.15 (reference :135-145)."""
    import subprocess
    process = subprocess.Popen(rapsearch + " -h", shell=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    output, error = process.communicate()
    if os.path.isdir(rapsearch) or len(error.decode().split("\n")) < 2:
        sys.exit("Problem executing rapsearch2: '%s'" % rapsearch)
    if error.decode().split("\n")[1] != "rapsearch v2.15: Fast protein similarity search tool for short reads":
        sys.exit("Incorrect version of rapsearch2 detected:'%s\nMicrobeCensus requires rapsearch v2.15" % rapsearch)


def _rapdb_for_external_search():
    """The marker database in RAPsearch2's on-disk format (what `prerapsearch -d markers.faa -n rapdb_2.15` writes), produced once
    per user by the library's own writer (mc_rapdb_write) from the packaged markers.  It lives in a directory only this user can
    write (~/.cache/microbecensus_amd, mode 0700); the pair of files is written into a fresh directory and renamed into place as a
    whole, and a database found there is checked against the packaged markers (mc_rapdb_verify) before it is used."""
    import hashlib
    import shutil
    import tempfile
    from . import _native
    names, seqs = _native.load_markers()
    tag = hashlib.md5(("".join(names) + "".join(seqs)).encode()).hexdigest()[:16]
    root = os.path.join(os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache"), "microbecensus_amd")
    os.makedirs(root, mode=0o700, exist_ok=True)
    st = os.stat(root)
    if st.st_uid != os.getuid() or (st.st_mode & 0o022):
        root = tempfile.mkdtemp(prefix="microbecensus_amd_")          # somebody else's directory: a private one for this run
    d = os.path.join(root, "rapdb_2.15_" + tag)
    path = os.path.join(d, "rapdb_2.15")
    if os.path.isfile(path) and os.path.isfile(path + ".info") and _native.rapdb_verify(path, names, seqs) == 0:
        return path
    tmp = tempfile.mkdtemp(prefix="rapdb_", dir=root)
    _native.rapdb_write(names, seqs, os.path.join(tmp, "rapdb_2.15"))
    if os.path.isdir(d):
        shutil.rmtree(d, ignore_errors=True)
    try:
        os.rename(tmp, d)
    except OSError:                                                   # another process of this user was faster
        if not (os.path.isfile(path) and os.path.isfile(path + ".info")):
            return os.path.join(tmp, "rapdb_2.15")
        shutil.rmtree(tmp, ignore_errors=True)
    return path


def _search_seqs_external(args, paths):
    """search_seqs exactly as the reference runs it (:369-389), with the executable given by -r / args['rapsearch']."""
    import subprocess
    command = "%s -q %s -d %s -o %s -z %s -e 1 -t n -p f -b 0" % (paths["rapsearch"], paths["tempfile"], _rapdb_for_external_search(), paths["tempfile"], args["threads"])
    process = subprocess.Popen(command, shell=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    output, error = process.communicate()
    if process.returncode != 0:
        clean_up(paths)
        sys.exit("\nDatabase search has exited with the following error:\n%s" % error)
    _run_cache.setdefault(paths["tempfile"], {}).pop("best", None)       # classify_reads parses the m8 file, like the reference
    if args["verbose"]:
        with open(paths["tempfile"] + ".m8") as f_in:
            print("\t%s reads hit marker proteins" % len(set(line.split()[0] for line in f_in if line[0] != "#")))


def check_paths(paths):
    for p in paths.values():
        if not (os.path.isfile(p) or os.path.isdir(p)):
            sys.exit("Path to file/dir not found: %s" % p)


# ----------------------------------------------------------------------------------------------
# FASTA / FASTQ reading (reference parse_seqs :294-325, a readfq derivative)
# ----------------------------------------------------------------------------------------------
class Sequence:
    _comp = {"A": "T", "T": "A", "G": "C", "C": "G", "N": "N"}

    def __init__(self, name, seq, quality=None):
        self.id, self.seq, self.quality = name, seq, quality

    def reverse_complement(self):
        return "".join(self._comp[b] for b in reversed(self.seq))   # KeyError on anything but ACGTN, as the reference

    def phred(self, offset):
        return [ord(c) - offset for c in self.quality]


def parse_seqs(fp):
    """Yields Sequence records from a FASTA/FASTQ text stream.  Every line loses exactly its last character
    (the newline) as in the reference; a '+' line after the sequence switches to quality mode, and quality
    lines are consumed until they cover the sequence length."""
    pending = None
    it = iter(fp)
    while True:
        if pending is None:
            for line in it:
                if line[:1] in (">", "@"):
                    pending = line[:-1]
                    break
            if not pending:                      # end of file, or a lone '>' / '@' without newline as the last line
                return
        name = pending[1:].partition(" ")[0]
        pending, chunks = None, []
        for line in it:
            if line[:1] in ("@", "+", ">"):
                pending = line[:-1]
                break
            chunks.append(line[:-1])
        seq = "".join(chunks)
        if not pending or pending[0] != "+":     # (a lone '+', '>' or '@' without newline as the last line leaves '')
            yield Sequence(name, seq)
            if not pending:
                return
            continue
        quals, got, done = [], 0, False
        for line in it:
            quals.append(line[:-1])
            got += len(line) - 1
            if got >= len(seq):
                done = True
                break
        if done:
            pending = None
            yield Sequence(name, seq, "".join(quals))
        else:
            yield Sequence(name, seq)
            return


read_seqfile_records = parse_seqs


def auto_detect_file_type(seqfile):
    with open_file(seqfile) as f_in:
        for line in f_in:
            if line[0] == ">":
                return "fasta"
            if line[0] == "@":
                return "fastq"
            sys.exit("Filetype [fasta, fastq] of %s could not be recognized" % seqfile)


def auto_detect_quality_offset(seqfile):
    """32 or 64, literally (reference :175-187: the value is later subtracted from ord(char))."""
    if not os.environ.get("MCENSUS_PYTHON_READER"):   # the same walk by the native parser (a file whose qualities all lie in ':'..'J' is walked to its end)
        try:
            from . import _native
            v = _native.quality_offset(seqfile)
            if v is not None:
                return v
        except Exception:
            pass
    low = set("""!"#$%&'()*+,-./0123456789""")
    high = set("""KLMNOPQRSTUVWXYZ[\\]^_`abcdefghijklmnopqrstuvwxyz{|}~""")
    with open_file(seqfile) as f_in:
        for rec in parse_seqs(f_in):
            for ch in rec.quality:
                if ch in low:
                    return 32
                if ch in high:
                    return 64
    return 32


def auto_detect_read_length(seqfile, file_type):
    lengths = []
    try:
        with open_file(seqfile) as f_in:
            for i, rec in enumerate(parse_seqs(f_in)):
                if i == 10000:
                    break
                lengths.append(len(rec.seq))
    except Exception:
        sys.exit("Could not detect read length of: %s\nThis may be due to an invalid format\nTry specifying it with -l" % seqfile)
    med = int(median(lengths))
    if med < VALID_READ_LENGTHS[0]:
        sys.exit("Median read length is %s. Cannot compute AGS using reads shorter than 50 bp." % med)
    best = VALID_READ_LENGTHS[-1]
    for i, L in enumerate(VALID_READ_LENGTHS):
        if L > med:
            best = VALID_READ_LENGTHS[i - 1]
            break
    return best


def impute_missing_args(args):
    for key, val in (("verbose", False), ("outfile", None), ("nreads", 1000000), ("threads", 1), ("filter_dups", False),
                     ("keep_tmp", False), ("mean_quality", -5), ("min_quality", -5), ("max_unknown", 100)):
        if key not in args:
            args[key] = val
    args["file_type"] = auto_detect_file_type(args["seqfiles"][0])
    if args["file_type"] == "fastq":
        args["quality_offset"] = auto_detect_quality_offset(args["seqfiles"][0])
    if "read_length" not in args or args["read_length"] is None:
        args["read_length"] = auto_detect_read_length(args["seqfiles"][0], args["file_type"])


def check_input(args):
    for seqfile in args["seqfiles"]:
        if not os.path.isfile(seqfile):
            sys.exit("Input file %s not found" % seqfile)


def check_arguments(args):
    if args["file_type"] == "fasta" and any([args["min_quality"] > -5, args["mean_quality"] > -5]):
        sys.exit("Quality filtering options are only available for FASTQ files")
    if args["threads"] < 1:
        sys.exit("Invalid number of threads: %s\nMust be a positive integer." % args["threads"])
    if args["nreads"] is not None and args["nreads"] < 1:
        sys.exit("Invalid number of reads: %s\nMust be a positive integer." % args["nreads"])


def print_copyright():
    print("\nMicrobeCensus - estimation of average genome size from shotgun sequence data")
    print("version %s; github.com/snayfach/MicrobeCensus (MI355X-native search path)" % __version__)
    print("Freely distributed under the GNU General Public License (GPLv3)\n")


def print_parameters(args):
    fq = args["file_type"] == "fastq"
    print("=============Parameters==============")
    print("Input metagenome: %s" % args["seqfiles"])
    print("Output file: %s" % args["outfile"])
    print("Reads trimmed to: %s bp" % args["read_length"])
    print("Maximum reads sampled: %s" % args["nreads"])
    print("Threads to use for db search: %s" % args["threads"])
    print("Minimum base-level quality score: %s" % (args["min_quality"] if fq else "NA"))
    print("Minimum read-level quality score: %s" % (args["mean_quality"] if fq else "NA"))
    print("Maximum percent unknown bases/read: %s" % args["max_unknown"])
    print("Filter duplicate reads: %s" % args["filter_dups"])
    print("Keep temporary files: %s\n" % args["keep_tmp"])


def quality_filter(rec, args):
    """True when the read fails QC; only the first read_length bases/qualities count (reference :265-279)."""
    L = args["read_length"]
    head = rec.seq[0:L]
    if 100 * head.count("N") / float(len(head)) > args["max_unknown"]:
        return True
    if args["file_type"] == "fastq":
        q = rec.phred(args["quality_offset"])[0:L]
        if mean(q) < args["mean_quality"]:
            return True
        if min(q) < args["min_quality"]:
            return True
    return False


# ----------------------------------------------------------------------------------------------
# stages
# ----------------------------------------------------------------------------------------------
def _process_seqfile_py(args, paths):
    """The sampler in Python (bz2 inputs; also the readable statement of the rules the native reader follows)."""
    L, nreads = args["read_length"], args["nreads"]
    kept, seen = [], set()
    dups = too_short = low_qual = 0
    with open(paths["tempfile"], "w") as out:
        for seqfile in args["seqfiles"]:
            for rec in parse_seqs(open_file(seqfile)):
                if len(rec.seq) < L:
                    too_short += 1
                    continue
                if args["filter_dups"] and (rec.seq in seen or rec.reverse_complement() in seen):
                    dups += 1
                    continue
                if quality_filter(rec, args):
                    low_qual += 1
                    continue
                out.write(">%d\n%s\n" % (len(kept), rec.seq[0:L]))
                kept.append(rec.seq[0:L])
                if args["filter_dups"]:
                    seen.add(rec.seq)
                if len(kept) == nreads:
                    break
            if len(kept) == nreads:
                break
    return (_pack_reads(kept, L) if kept else np.zeros((0, L), np.uint8)), {"sampled": len(kept), "too_short": too_short, "low_qual": low_qual, "dups": dups}


def _native_reader_usable(args):
    return not args.get("python_reader")          # (args['python_reader']: the sampler in Python, the readable statement of the rules)


def process_seqfile(args, paths):
    """Head-take sampler: files in order, records in order; too short -> skip; duplicate (full sequence or its
    reverse complement, checked before QC) -> skip; QC fail -> skip; else keep seq[:L]; stop at nreads.
    Runs in the native reader of libmcensus_hip.so (mc_reader_*, csrc/mc_reader.cpp); .bz2 inputs in Python."""
    if args["verbose"]:
        print("====Estimating Average Genome Size====")
        print("Sampling & trimming reads...")
    L = args["read_length"]
    if _native_reader_usable(args):
        from . import _native
        try:
            nreads = args["nreads"] if args["nreads"] is not None else (1 << 63) - 1      # None = no cap, as in the reference (read_id == None is never true)
            reads, st = _native.sample_reads(args["seqfiles"], L, nreads, args["file_type"] == "fastq", args.get("quality_offset") or 0,
                                             args["min_quality"], args["mean_quality"], args["max_unknown"], args["filter_dups"], paths["tempfile"])
        except _native.ReferenceError_ as e:      # the reference raises here; run_pipeline prints it and returns None
            raise Exception(str(e))
    else:
        reads, st = _process_seqfile_py(args, paths)
    if st.get("exhausted"):                        # the sampler saw every record: count_bases() need not read the files again
        _bases_cache[tuple(args["seqfiles"])] = st["bases"]
    if st["sampled"] == 0:
        clean_up(paths)
        sys.exit("\nError! No reads remaining after filtering!")
    args["sampled_reads"] = st["sampled"]
    _run_cache[paths["tempfile"]] = {"reads": reads}
    if args["verbose"]:
        print("\t%s reads shorter than %s bp and skipped" % (st["too_short"], L))
        print("\t%s low quality reads found and skipped" % st["low_qual"])
        print("\t%s duplicate reads found and skipped" % st["dups"])
        print("\t%s reads sampled from seqfile" % st["sampled"])


def _sample_search_classify(args, paths):
    """process_seqfile + search_seqs + the device half of classify_reads in one native call (mc_search_files): the sampler
    reads beside the search, batch by batch; no temp FASTA and no m8 text are written (run_pipeline deletes both unseen)."""
    from . import _native
    L = args["read_length"]
    if args["verbose"]:
        print("====Estimating Average Genome Size====")
        print("Sampling & trimming reads...")
    model = _model()
    fams = model["families"]
    rd = _native.Reader(args["seqfiles"], L, args["nreads"], args["file_type"] == "fastq", args.get("quality_offset") or 0,
                        args["min_quality"], args["mean_quality"], args["max_unknown"], args["filter_dups"])
    try:
        try:
            devs = _devices_for(args)
            engs = _engines_on(devs)
            for eng in engs:
                eng.set_run(L, model["pars"][str(L)], fams)
            # (the count of reads with m8 rows is only printed when verbose: without it only the reads that can be classified are ranked)
            if len(engs) == 1:
                rows, best = engs[0].search_files(rd, keep_rows=False, best_only=not args["verbose"])
            else:       # one sampler, batches of 2 M accepted reads dealt to the GPUs as they ask for them (mc_search_files_multi): same best hits
                best = _native.search_files_multi(engs, rd, keep_rows=False, best_only=not args["verbose"])
        except _native.ReferenceError_ as e:       # the reference raises here; run_pipeline prints it and returns None
            raise Exception(str(e))
        except RuntimeError as error:
            clean_up(paths)
            sys.exit("\nDatabase search has exited with the following error:\n%s" % error)
        st = rd.stats()
        args["_sampler_seconds"] = rd.times()      # (where the host side of the call went: bench.py's e2e legs report it)
        hit_reads = sum(e.stats()["reads_with_rows"] for e in engs)
    finally:
        rd.close()
    if st.get("exhausted"):
        _bases_cache[tuple(args["seqfiles"])] = st["bases"]
    if st["sampled"] == 0:
        clean_up(paths)
        sys.exit("\nError! No reads remaining after filtering!")
    args["sampled_reads"] = st["sampled"]
    _run_cache[paths["tempfile"]] = {"reads": None, "best": best, "families": fams}
    if args["verbose"]:
        print("\t%s reads shorter than %s bp and skipped" % (st["too_short"], L))
        print("\t%s low quality reads found and skipped" % st["low_qual"])
        print("\t%s duplicate reads found and skipped" % st["dups"])
        print("\t%s reads sampled from seqfile" % st["sampled"])
        print("Searching reads against marker proteins...")
        print("\t%s reads hit marker proteins" % hit_reads)


def _cap_host_threads(threads):
    """args['threads'] is the reference's rapsearch -z (:375); here the search runs on the GPU and the host threads are the
    native sampler's workers: an explicit value caps them, without one the CPUs the process may use (its cgroup quota; up to 32)."""
    try:
        from . import _native
        _native.load_library().mc_set_host_threads(int(threads) if threads and int(threads) > 0 else 0)
    except Exception:
        pass


def _engine(device):
    from . import _native
    if device not in _engines:
        _engines[device] = _native.Engine(device=device)
    return _engines[device]


STREAM_BATCH = 2000000      # accepted reads per batch of mc_search_files(_multi)


def _devices_for(args):
    """The GPUs a run uses.  args['devices']: a list of device indices, or 'all'; else args['device'] (default 0) when given; else
    every visible device - but never more than the library can keep busy: batches of 2 M accepted reads are dealt to the devices,
    so a run of n reads uses at most ceil(n / 2 M) of them (the reference's default -n of 1 - 2 M reads: one GPU).  The results do
    not depend on it (reads are independent; per-family sums are integers)."""
    from . import _native
    devs = args.get("devices")
    if devs is None and args.get("device") is not None:
        return [int(args["device"])]
    visible = max(1, _native.load_library().mc_device_count())
    if devs is None or devs == "all":
        devs = list(range(visible))
    devs = [int(d) for d in devs]
    n = args.get("nreads")
    if n is not None:
        devs = devs[: max(1, -(-int(n) // STREAM_BATCH))]
    return devs or [0]


def _engines_on(devs):
    """One engine per entry of devs (an index may repeat: several handles on one GPU), opened in parallel: mc_open builds the
    marker index on the host, a second of work per handle."""
    import threading
    from . import _native
    keys, seen = [], {}
    for d in devs:                                   # the second handle on device d is engine (d, 1) ...
        k = seen.get(d, 0)
        seen[d] = k + 1
        keys.append(d if k == 0 else (d, k))
    missing = [k for k in keys if k not in _engines]
    errs = []

    def make(k):
        try:
            _engines[k] = _native.Engine(device=k if isinstance(k, int) else k[0])
        except Exception as e:          # noqa: BLE001
            errs.append(e)
    ths = [threading.Thread(target=make, args=(k,)) for k in missing]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    if errs:
        raise RuntimeError(str(errs[0]))
    return [_engines[k] for k in keys]


def _pack_reads(seqs, L):
    """(n, L) uint8; characters outside latin-1 cannot be bases, map them to '?'."""
    blob = "".join(seqs).encode("latin-1", "replace")
    return np.frombuffer(blob, dtype=np.uint8).reshape(len(seqs), L)


def search_seqs(args, paths):
    """Translated search of the trimmed reads against the marker proteins on the GPU; leaves
    `<tempfile>.m8` (RAPsearch2 m8 format) behind like the reference does."""
    if args["verbose"]:
        print("Searching reads against marker proteins...")
    if paths.get("rapsearch"):
        return _search_seqs_external(args, paths)
    L = args["read_length"]
    cache = _run_cache.get(paths["tempfile"])
    if cache is None or "reads" not in cache:
        seqs = [r.seq for r in parse_seqs(open(paths["tempfile"]))]
        cache = _run_cache[paths["tempfile"]] = {"reads": _pack_reads(seqs, L)}
    model = _model()
    fams = model["families"]
    try:
        eng = _engine(args.get("device", 0) or 0)
        eng.set_run(L, model["pars"][str(L)], fams)
        rows, best = eng.search(cache["reads"])
        with open(paths["tempfile"] + ".m8", "w") as f:
            f.write("# microbecensus_amd %s: RAPsearch2-compatible m8 written by the MI355X search path\n" % __version__)
            f.write("# Job submitted: reads=%d trimmed to %d bp\n# Query : %s\n# Subject : %s\n" % (len(cache["reads"]), L, paths["tempfile"], paths["db"]))
            f.write("# Fields: Query\tSubject\tidentity\taln-len\tmismatch\tgap-openings\tq.start\tq.end\ts.start\ts.end\tlog(e-value)\tbit-score\n")
        eng.write_m8(paths["tempfile"] + ".m8", append=True)
    except Exception as error:
        clean_up(paths)
        sys.exit("\nDatabase search has exited with the following error:\n%s" % error)
    cache["best"], cache["families"], cache["rows"] = best, fams, rows
    if args["verbose"]:
        print("\t%s reads hit marker proteins" % len(np.unique(rows["query"])))


def parse_rapsearch(m8):
    """Records of an m8 file as dicts of 2 strings and 10 floats (reference :391-398)."""
    names = ("query", "target", "pid", "aln", "mis", "gaps", "qstart", "qend", "tstart", "tend", "evalue", "score")
    with open(m8) as f_in:
        for line in f_in:
            if line[0] == "#":
                continue
            vals = line.rstrip().split()
            yield dict(zip(names, vals[:2] + [float(v) for v in vals[2:12]]))


def alignment_coverage(r):
    """aln / (largest alignment the read could have had at this position), reference :400-418."""
    qlen = float(r["query_len"]) / 3
    qs, qe = sorted([r["qstart"], r["qend"]])
    frame = qs % 3 if qs % 3 in [1, 2] else 3
    q_start = (qs + 3 - frame) / 3
    q_stop = (qe + 1 - frame) / 3
    t_start, t_stop = sorted([r["tstart"] + 1, r["tend"] + 1])
    return r["aln"] / (min(q_start - 1, t_start - 1) + r["aln"] + min(qlen - q_stop, r["target_len"] - t_stop))


def alignment_filter(r, optpars):
    p = optpars[r["target_fam"]]
    return alignment_coverage(r) < p["min_cov"] or r["score"] < p["min_score"] or r["pid"] > p["max_aaid"]


def _classify_m8_file(args, paths):
    """The reference's own classification of an m8 file (used when the file did not come from this process,
    and by the tests as the checker of the device classification)."""
    from . import _native
    optpars = find_opt_pars(None, args["read_length"])
    model = _model()
    names, seqs = _native.load_markers(paths.get("db"))
    fam_of = {n: model["families"][f] for n, f in zip(names, model["marker_family"])}
    len_of = {n: float(len(s)) for n, s in zip(names, seqs)}
    best = {}
    for r in parse_rapsearch(paths["tempfile"] + ".m8"):
        r["query_len"], r["target_fam"], r["target_len"] = args["read_length"], fam_of[r["target"]], len_of[r["target"]]
        if alignment_filter(r, optpars):
            continue
        if r["query"] not in best or best[r["query"]][-1] < r["score"]:
            best[r["query"]] = [r["target_fam"], r["aln"], r["aln"] / r["target_len"], r["score"]]
    return best


class _BestHits(dict):
    """classify_reads' result {read_id(str): [family, aln, aln/target_len, bit score]} backed by the device's best-hit array.  The dict
    itself (a Python object per classified read: 0.35 s for the 126,000 of a 20 M-read library) is only built when somebody looks
    into it; run_pipeline does not - aggregate_hits() sums straight from the array, in the dict's order."""

    def __init__(self, best, fams):
        dict.__init__(self)
        self._best, self._fams, self._built = best, fams, False

    def _build(self):
        if not self._built:
            self._built = True
            b, fams = self._best, self._fams
            aln = b["aln"].astype(np.float64)
            cov = aln / b["target_len"].astype(np.float64)
            dict.update(self, {str(r): [fams[f], a, c, s] for r, f, a, c, s in zip(b["read"].tolist(), b["family"].tolist(), aln.tolist(), cov.tolist(), b["bits"].astype(np.float64).tolist())})
        return self

    def __len__(self):
        return len(self._best) if not self._built else dict.__len__(self)

    def __bool__(self):
        return len(self) > 0


def _lazy(name):
    def f(self, *a, **k):
        return getattr(dict, name)(self._build(), *a, **k)
    f.__name__ = name
    return f


for _n in ("__getitem__", "__iter__", "__contains__", "__eq__", "__ne__", "__repr__", "__setitem__", "__delitem__", "__reduce__", "__reduce_ex__", "__or__", "__ror__", "__ior__",
           "keys", "values", "items", "get", "pop", "popitem", "setdefault", "update", "copy", "clear"):
    setattr(_BestHits, _n, _lazy(_n))
_BestHits.__hash__ = None


def classify_reads(args, paths):
    """{read_id(str): [family, aln, aln/target_len, bit score]} of the best passing hit of every read."""
    if args["verbose"]:
        print("Filtering hits...")
    cache = _run_cache.get(paths["tempfile"], {})
    if cache.get("best") is not None:
        best_hits = _BestHits(cache["best"], cache["families"])      # ascending read id = the order the reference meets them in the m8
    else:
        best_hits = _classify_m8_file(args, paths)
    if len(best_hits) == 0:
        clean_up(paths)
        sys.exit("\nError: No hits to marker proteins - cannot estimate genome size! Rerun program with more reads.")
    if args["verbose"]:
        print("\t%s reads assigned to a marker protein" % len(best_hits))
    return best_hits


def aggregate_hits(args, paths, best_hits):
    """Per family: number of hits, summed aln/target_len, or summed aln, as pars.map's aln_stat says."""
    optpars = find_opt_pars(None, args["read_length"])
    if isinstance(best_hits, _BestHits) and not best_hits._built:
        # the same sums from the array: families in the order of their first hit (the dict's insertion order - estimate's weighted
        # sum runs over it), every family's increments added one after the other in read order (np.add.accumulate is that running sum)
        b, fams = best_hits._best, best_hits._fams
        fam = b["family"]
        aln = b["aln"].astype(np.float64)
        cov = aln / b["target_len"].astype(np.float64)
        present, first = np.unique(fam, return_index=True)
        agg = {}
        for f in present[np.argsort(first)].tolist():
            name = fams[f]
            stat = optpars[name]["aln_stat"]
            sel = fam == f
            inc = np.ones(int(sel.sum())) if stat == "hits" else cov[sel] if stat == "cov" else aln[sel]
            agg[name] = float(np.add.accumulate(inc)[-1])
        return agg
    agg = {}
    for fam, aln, cov, score in best_hits.values():
        stat = optpars[fam]["aln_stat"]
        inc = 1.0 if stat == "hits" else cov if stat == "cov" else aln
        agg[fam] = agg[fam] + inc if fam in agg else inc
    return agg


def estimate_average_genome_size(args, paths, agg_hits):
    """AGS_j = coefficient_j / (hits_j / sampled bp); drop |AGS_j - median| >= 1.48 MAD; weighted mean."""
    if args["verbose"]:
        print("Computing average genome size...")
    model = _model()
    L = str(args["read_length"])
    estimates = {}
    for fam, hits in agg_hits.items():
        rate = hits / (args["sampled_reads"] * args["read_length"])
        if rate == 0:
            continue
        estimates[fam] = model["coefficients"]["_".join([L, fam])] / rate
    spread = mad(list(estimates.values()))
    centre = median(list(estimates.values()))
    total = wsum = 0
    for fam, est in estimates.items():
        if abs(est - centre) >= spread:
            continue
        w = model["weights"]["_".join([L, fam])]
        total += est * w
        wsum += w
    est_ags = total / wsum
    if args["verbose"]:
        print("\t%s bp" % str(round(est_ags, 2)))
    return est_ags


def report_results(args, est_ags, count_bases):
    with open(args["outfile"], "w") as out:
        out.write("Parameters\n")
        for key, val in (("metagenome", ",".join(args["seqfiles"])), ("reads_sampled", args["sampled_reads"]), ("trimmed_length", args["read_length"]),
                         ("min_quality", args["min_quality"]), ("mean_quality", args["mean_quality"]), ("filter_dups", args["filter_dups"]),
                         ("max_unknown", args["max_unknown"])):
            out.write("%s:\t%s\n" % (key, val))
        out.write("\nResults\n")
        out.write("%s:\t%s\n" % ("average_genome_size", est_ags))
        if count_bases:
            out.write("%s:\t%s\n" % ("total_bases", count_bases))
            out.write("%s:\t%s\n" % ("genome_equivalents", count_bases / est_ags))


def clean_up(paths):
    base = paths["tempfile"]
    for ext in ("", ".m8", ".aln"):
        for i in range(20):
            f = "%s%s.tmp%s" % (base, ext, i)
            if os.path.isfile(f):
                os.remove(f)
        if os.path.isfile(base + ext):
            os.remove(base + ext)
    _run_cache.pop(base, None)


def read_seqfile(infile):
    for rec in parse_seqs(infile):
        yield rec.id, rec.seq, rec.quality


def count_bases(args):
    if args["verbose"]:
        print("Computing number of genome equivalents...")
    cached = _bases_cache.pop(tuple(args["seqfiles"]), None)
    if cached is not None:
        return cached
    if _native_reader_usable(args):
        from . import _native
        try:
            return _native.count_bases(args["seqfiles"])
        except _native.ReferenceError_ as e:
            raise Exception(str(e))
    total = 0
    for inpath in args["seqfiles"]:
        with open_file(inpath) as infile:
            total += sum(len(rec.seq) for rec in parse_seqs(infile))
    return total


def run_pipeline(args):
    if "verbose" in args and args["verbose"]:
        print_copyright()
    check_os()
    paths = get_relative_paths(args)
    check_paths(paths)
    try:
        check_input(args)
        _cap_host_threads(args.get("threads"))     # an explicit args['threads'] (-t) caps the sampler's worker threads
        impute_missing_args(args)
        check_arguments(args)
        if args["verbose"]:
            print_parameters(args)
        if paths.get("rapsearch"):
            check_rapsearch(paths["rapsearch"])
        if _native_reader_usable(args) and not args.get("keep_tmp") and not paths.get("rapsearch"):
            _sample_search_classify(args, paths)
        else:                                      # stage by stage, with the temp FASTA and the m8 file the reference leaves behind
            process_seqfile(args, paths)
            search_seqs(args, paths)
        best_hits = classify_reads(args, paths)
        agg_hits = aggregate_hits(args, paths, best_hits)
        clean_up(paths)
        est_ags = estimate_average_genome_size(args, paths, agg_hits)
        return est_ags, args
    except Exception as error:     # the reference prints and swallows everything but SystemExit
        print(error)
        clean_up(paths)
