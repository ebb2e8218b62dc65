// mc_reader.cpp - native read sampler: the host stage in front of the search (include/mcensus.h, mc_reader_*).
//
// Replaces, with identical results, the Python stages of the reference that feed the hot path
//   open_file            /root/reference/microbe_census/microbe_census.py:47-59   (plain, .gz and .bz2)
//   parse_seqs           :294-325   readfq-style FASTA/FASTQ records, with its quirks (below)
//   quality_filter       :265-279
//   process_seqfile      :328-367   head-take sampler: files in order, records in order
//   count_bases          :573-584   second pass over every file
// and hands the accepted reads over as packed read_len-byte rows, which is what mc_search() / mc_upload() take.
//
// Quirks of the reference reproduced here:
//   * text mode with universal newlines: "\r\n" and a lone "\r" end a line like "\n";
//   * every line loses exactly its last character - the newline - so the last line of a file that does not end in a
//     newline loses its last real character; a lone '+', '>' or '@' as such a last line becomes '' and ends the parse;
//   * sequence lines run until a line that starts with '@', '+' or '>'; after '+', quality lines are consumed until they
//     cover the sequence length; a file that ends inside the qualities yields the record without qualities and stops;
//   * a read is too short when len(seq) < L; duplicates (the untrimmed sequence or its reverse complement already
//     accepted) are tested BEFORE the quality filter and only when requested; only accepted reads enter the set;
//     reverse_complement knows ACGTN only - any other character is an error (KeyError in the reference);
//   * QC looks at the first L bases / qualities: 100*count('N')/L > max_unknown, mean(q) < mean_quality, min(q) < min_quality
//     with q = ord(c) - quality_offset, in IEEE double like numpy;
//   * a truncated or corrupt compressed stream is an error (gzip.open raises EOFError) - but only if the sampler gets there;
//   * the codec follows the file NAME (.gz, .bz2, anything else is plain text), as open_file does: a *.gz file that does not hold
//     gzip data is an error (BadGzipFile), gzip data under another name is read as the bytes it is.
//
// How it is made fast (the sampler is sequential by definition: head-take, first occurrence wins):
//   * the byte stream is taken in regions of whole lines (plain files: slices of one mmap, nothing is copied; compressed
//     files: a producer thread inflates ahead while the region in hand is parsed);
//   * a region is cut into pieces at GUESSED record starts ('@' line whose second next line starts with '+', or a '>' line);
//     the worker threads run the exact state machine of parse_seqs on the pieces, each from its guess up to the next one, and
//     evaluate their records (length, N count, quality sum / minimum, 64-bit hashes of the sequence and of its reverse
//     complement).  Then the pieces are stitched in order: a piece counts only if its predecessor ended exactly where it
//     started - otherwise the predecessor's parser is continued sequentially until it meets a later piece's start (never
//     happens on well-formed files, keeps odd files exact);
//   * the merge walks the record descriptors in file order (too short / duplicate / low quality / accept, duplicates against
//     an exact set keyed by the precomputed hashes) and assigns output slots; the workers copy the accepted reads.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <thread>
#include <vector>

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include "../../include/mcensus.h"
#include "mc_pgzip.h"
#include "mc_pbzip2.h"
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

namespace {

thread_local std::string r_err;
thread_local int64_t t_range_lo = -1, t_range_hi = -1;   // mc_reader_open_range: the byte window of the ONE plain file the reader is opened on (-1: the whole file)
thread_local int64_t t_bz_b0 = -1, t_bz_b1 = -1; thread_local int t_bz_kind = 0;   // mc_reader_open_bz2_part: the blocks [b0, b1) of the ONE .bz2 file the reader is opened on, and what a record of the file starts with
// mc_reader_open_gz_part: the chunks [k0, k1) of the ONE .gz file the reader is opened on; what it learns from the owner of the slice in front
// and tells the owner of the next one (mc_reader_gz_provide / mc_reader_gz_end_state), and the member CRCs it checks at the end (mc_reader_gz_finish)
struct GzPartCtx {
    std::mutex mu; std::condition_variable cv;
    bool in_ready = false, in_fail = false, out_ready = false, out_fail = false;
    mcgz::ParallelGz::SliceState in, out;
    std::vector<mcgz::ParallelGz::SegRec> segs; size_t nsegs_own = 0;
};
thread_local int64_t t_gz_k0 = -1, t_gz_k1 = -1, t_gz_chunk = 0; thread_local int t_gz_kind = 0; thread_local GzPartCtx *t_gz_ctx = nullptr;
thread_local bool t_peek = false;   // the caller will most likely stop after a few records (mc_quality_offset): no parallel inflate, small first regions

// ------------------------------------------------------------------------------------------------------------------
// worker pool: run(n, f) calls f(i) for i in [0, n) on the pool's threads and the caller
// ------------------------------------------------------------------------------------------------------------------
struct Pool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv, cv_done;
    const std::function<void(int)> *job = nullptr;
    int njobs = 0, busy = 0;
    std::atomic<int> next{0};
    uint64_t gen = 0;
    bool stop = false;
    explicit Pool(int nthreads) { for (int i = 1; i < nthreads; i++) th.emplace_back([this] { loop(); }); }
    ~Pool()
    {
        { std::unique_lock<std::mutex> lk(mu); stop = true; cv.notify_all(); }
        for (auto &t : th) t.join();
    }
    int size() const { return (int)th.size() + 1; }
    void loop()
    {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(int)> *f;
            int n;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || gen != seen; });
                if (stop) return;
                seen = gen; f = job; n = njobs;
                if (!f) continue;
                busy++;
            }
            for (int i; (i = next.fetch_add(1)) < n;) (*f)(i);
            { std::unique_lock<std::mutex> lk(mu); if (--busy == 0) cv_done.notify_all(); }
        }
    }
    void run(int n, const std::function<void(int)> &f)
    {
        if (n <= 0) return;
        if (th.empty() || n == 1) { for (int i = 0; i < n; i++) f(i); return; }
        {
            std::unique_lock<std::mutex> lk(mu);
            job = &f; njobs = n; next.store(0); busy = 1; gen++;
            cv.notify_all();
        }
        for (int i; (i = next.fetch_add(1)) < n;) f(i);
        std::unique_lock<std::mutex> lk(mu);
        busy--;
        cv_done.wait(lk, [&] { return busy == 0; });
        job = nullptr;                                   // (a worker that wakes up late finds no job and goes back to sleep)
    }
};

// ------------------------------------------------------------------------------------------------------------------
// byte stream of one file: a window of contiguous bytes that can be extended at its end and consumed at its front
// ------------------------------------------------------------------------------------------------------------------
bool has_ext(const char *path, const char *ext) { size_t n = strlen(path), m = strlen(ext); return n >= m && strcmp(path + n - m, ext) == 0; }

// libbz2 is on every machine that runs Python's bz2 module, but its header is not in this image: the three entry points of
// its low-level interface are bound at run time (bz_stream as bzlib.h 1.0 declares it).  The low-level interface, not
// BZ2_bzread: a .bz2 file may hold several streams back to back (pbzip2 and lbzip2 write such files, `cat a.bz2 b.bz2` makes
// one) and Python's bz2 module reads all of them, while BZ2_bzread stops at the end of the first.
struct BzStream {
    char *next_in; unsigned int avail_in, total_in_lo32, total_in_hi32;
    char *next_out; unsigned int avail_out, total_out_lo32, total_out_hi32;
    void *state;
    void *(*bzalloc)(void *, int, int); void (*bzfree)(void *, void *); void *opaque;
};
struct Bz2Api {
    void *lib = nullptr;
    int (*init)(BzStream *, int, int) = nullptr;
    int (*decompress)(BzStream *) = nullptr;
    int (*end)(BzStream *) = nullptr;
    bool load()
    {
        if (lib) return init != nullptr;
        for (const char *n : {"libbz2.so.1.0", "libbz2.so.1", "libbz2.so"}) { lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (lib) break; }
        if (!lib) return false;
        init = (int (*)(BzStream *, int, int))dlsym(lib, "BZ2_bzDecompressInit");
        decompress = (int (*)(BzStream *))dlsym(lib, "BZ2_bzDecompress");
        end = (int (*)(BzStream *))dlsym(lib, "BZ2_bzDecompressEnd");
        if (!(init && decompress && end)) { init = nullptr; return false; }
        return true;
    }
};
Bz2Api g_bz2;
std::mutex g_bz2_mu;

// A .bz2 file as a sequence of bzip2 streams.  read() fills the buffer like a file read: < n only at the end of the data;
// bad = the file ends inside a stream or holds damaged data (Python: EOFError / OSError).  Bytes behind the last complete stream
// that do not start another one are ignored, as Python's bz2 module ignores them.
struct Bz2File {
    int fd = -1;
    BzStream z{};
    bool live = false, eof_in = false, fresh = true;               // fresh: nothing of the current stream has been decoded yet
    std::vector<char> in;
    bool open(const char *path) { fd = ::open(path, O_RDONLY); in.resize(1 << 20); return fd >= 0; }
    // ... at a byte offset of the file, behind streams somebody else has decoded (got_any: there were some): mc_pbzip2.h hands over here
    bool open_at(const char *path, size_t offset, bool got_any)
    {
        if (!open(path)) return false;
        if (lseek(fd, (off_t)offset, SEEK_SET) < 0) return false;
        got_any_stream = got_any;
        return true;
    }
    void close() { if (live) { g_bz2.end(&z); live = false; } if (fd >= 0) { ::close(fd); fd = -1; } }
    int read(uint8_t *dst, int n, bool *bad, std::string *msg)
    {
        int got = 0;
        *bad = false;
        while (got < n) {
            if (!live) {
                if (z.avail_in == 0 && eof_in) break;                  // clean end: between two streams
                memset(&z.state, 0, sizeof z.state); z.bzalloc = nullptr; z.bzfree = nullptr; z.opaque = nullptr;
                const unsigned keep_n = z.avail_in; char *keep_p = z.next_in;
                if (g_bz2.init(&z, 0, 0) != 0) { *bad = true; *msg = "BZ2_bzDecompressInit failed"; return got; }
                z.avail_in = keep_n; z.next_in = keep_p;
                live = true; fresh = true;
            }
            if (z.avail_in == 0 && !eof_in) {
                const ssize_t k = ::read(fd, in.data(), in.size());
                if (k < 0) { *bad = true; *msg = "read error"; return got; }
                if (k == 0) eof_in = true;
                z.next_in = in.data(); z.avail_in = (unsigned)k;
            }
            if (z.avail_in == 0 && eof_in) {                          // the input ends inside a stream
                if (fresh && got_any_stream) break;                    // (nothing but the end of the file behind the last stream)
                *bad = true; *msg = "compressed file ended before the end-of-stream marker was reached"; return got;
            }
            z.next_out = (char *)dst + got; z.avail_out = (unsigned)(n - got);
            const unsigned in0 = z.avail_in;
            const int rc = g_bz2.decompress(&z);
            got = n - (int)z.avail_out;
            if (in0 != z.avail_in) fresh = false;
            if (rc == 4 /* BZ_STREAM_END */) { g_bz2.end(&z); live = false; got_any_stream = true; continue; }
            if (rc != 0 /* BZ_OK */) {
                if (got_any_stream && first_block_of_stream()) { g_bz2.end(&z); live = false; z.avail_in = 0; eof_in = true; break; }   // trailing bytes that are no stream
                *bad = true; *msg = "invalid data stream"; return got;
            }
        }
        return got;
    }
    bool got_any_stream = false;
    // a decode error before the stream produced anything: the bytes behind the last stream were not a stream at all
    bool first_block_of_stream() const { return z.total_out_lo32 == 0 && z.total_out_hi32 == 0; }
};

int reader_threads();
int inflate_threads();
size_t guess_start(const uint8_t *base, size_t from, size_t e, int max_lines, int kind, bool skip_first = false);

struct Stream {
    enum { NBLK = 16, BLK = 1 << 22, BLK_PEEK = 1 << 16 };
    // plain file: one mapping; the window is a slice of it
    const uint8_t *map = nullptr; size_t map_n = 0;
    const uint8_t *vend = nullptr;              // end of the part of the mapping that is read (map + map_n, or the end of a byte window: mc_reader_open_range)
    bool ranged = false;
    // compressed stream: a producer thread inflates BLK-sized blocks into a ring; the window lives in `buf`
    gzFile gz = nullptr; Bz2File *bz = nullptr;
    mcgz::ParallelGz *pgz = nullptr; const uint8_t *gzmap = nullptr; size_t gzmap_n = 0;   // a regular .gz file: mapped and inflated in parallel (mc_pgzip.h)
    mcgz::SerialGz *sgz = nullptr;                                                          // ... or by one stream (one thread allowed; the quality-offset peek)
    mcbz::ParallelBz2 *pbz = nullptr; const uint8_t *bzmap = nullptr; size_t bzmap_n = 0;   // a regular .bz2 file: mapped, its well-formed streams decoded block by block in parallel (mc_pbzip2.h)
    std::string bz_path; uint64_t bz_skip = 0; bool bz_tail = false;                        // ... and what is left for the one-stream decoder
    std::vector<uint8_t> part_text;                                                         // mc_reader_open_bz2_part: the decoded text of the rank's blocks (and of a little behind them), which `map` then points at
    std::thread th;
    std::mutex mu; std::condition_variable cv;
    std::vector<uint8_t> ring[NBLK]; size_t ring_n[NBLK] = {};
    int head = 0, tail = 0, count = 0;
    bool prod_done = false, prod_err = false, stop = false;
    std::string prod_msg;
    std::vector<uint8_t> buf; size_t buf_at = 0;
    // the window
    const uint8_t *win = nullptr; size_t len = 0;
    bool at_end = false;                        // nothing can be added to the window any more
    bool failed = false;                        // ... because the stream is truncated / corrupt
    bool compressed = false;
    bool bad_gzip = false;                      // open() failed because a *.gz file does not hold gzip data
    int peek_blocks = 0;                        // the first blocks of the producer are small ones
    double t_prod_read = 0, t_prod_wait = 0, t_ext_wait = 0, t_ext_copy = 0;   // (MC_READER_TIMING)
    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

    ~Stream() { close(); }
    bool open(const char *path)
    {
        if (has_ext(path, ".bz2")) {
            std::unique_lock<std::mutex> lk(g_bz2_mu);
            if (!g_bz2.load()) { r_err = "cannot load libbz2 for " + std::string(path); return false; }
            bz_path = path;
            if (t_bz_b0 >= 0) return open_bz2_part(path);
            const int bzt = reader_threads();
            struct stat sb;
            if (bzt >= 2 && !t_peek && !getenv("MC_READER_SERIAL_BZ2") && stat(path, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size >= 14) {
                const int fd = ::open(path, O_RDONLY);
                void *m = fd >= 0 ? mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0) : MAP_FAILED;
                if (fd >= 0) ::close(fd);
                if (m != MAP_FAILED) {
                    bzmap = (const uint8_t *)m; bzmap_n = (size_t)sb.st_size;
                    mcbz::Api api;
                    api.init = (int (*)(mcbz::BzStreamT *, int, int))g_bz2.init; api.decompress = (int (*)(mcbz::BzStreamT *))g_bz2.decompress; api.end = (int (*)(mcbz::BzStreamT *))g_bz2.end;
                    pbz = new mcbz::ParallelBz2(bzmap, bzmap_n, api, std::min(bzt, 32));
                    if (!pbz->start()) { delete pbz; pbz = nullptr; munmap((void *)bzmap, bzmap_n); bzmap = nullptr; }   // (not even the first stream checks out: the one-stream decoder decides)
                }
            }
            if (!pbz) {
                bz = new Bz2File();
                if (!bz->open(path)) { delete bz; bz = nullptr; r_err = std::string("cannot open ") + path; return false; }
            }
            compressed = true;
        } else if (t_gz_k0 >= 0) {
            return open_gz_part(path);
        } else {
            int fd = ::open(path, O_RDONLY);
            if (fd < 0) { r_err = std::string("cannot open ") + path; return false; }
            unsigned char magic[2] = {0, 0};
            const ssize_t got = pread(fd, magic, 2, 0);
            struct stat sb;
            const bool reg = fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode);
            const bool is_gz = got == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
            // the codec follows the file NAME, as open_file does (reference :47-59): a *.gz that is not gzip is gzip.open's BadGzipFile
            if (has_ext(path, ".gz") && reg && got > 0 && !is_gz) { ::close(fd); r_err = std::string("BadGzipFile: Not a gzipped file (") + path + ")"; bad_gzip = true; return false; }
            const int gzt = reader_threads();
            if (has_ext(path, ".gz") && reg && is_gz && gzt >= 2 && !t_peek && !getenv("MC_READER_SERIAL_GZ")) {
                // several inflate workers beside the parser (one inflate stream tops out at ~0.5 GB/s of text); with one thread allowed, zlib's own reader below
                gzmap_n = (size_t)sb.st_size;
                void *m = mmap(nullptr, gzmap_n, PROT_READ, MAP_PRIVATE, fd, 0);
                ::close(fd);
                if (m == MAP_FAILED) { r_err = std::string("cannot map ") + path; return false; }
                madvise(m, gzmap_n, MADV_SEQUENTIAL);
                gzmap = (const uint8_t *)m;
                size_t chunk = (size_t)1 << 20;                          // compressed bytes per speculative chunk
                if (const char *v = getenv("MC_READER_GZ_CHUNK")) chunk = std::max<size_t>(4096, (size_t)atoll(v));   // (tests: many chunks on small files)
                pgz = new mcgz::ParallelGz(gzmap, gzmap_n, std::min(inflate_threads(), 32), chunk);
                if (!pgz->start()) {                                     // (a header it does not take - cut short, odd fields: zlib's reader decides)
                    delete pgz; pgz = nullptr; munmap(m, gzmap_n); gzmap = nullptr;
                    gz = gzopen(path, "rb");
                    if (!gz) { r_err = std::string("cannot open ") + path; return false; }
                    gzbuffer(gz, 1 << 20);
                }
                compressed = true;
            } else if (has_ext(path, ".gz") && reg && is_gz) {           // one thread allowed, or the quality-offset peek: one inflate stream, gzip.open's rules between the members
                gzmap_n = (size_t)sb.st_size;
                void *m = mmap(nullptr, gzmap_n, PROT_READ, MAP_PRIVATE, fd, 0);
                if (m == MAP_FAILED) { ::close(fd); r_err = std::string("cannot map ") + path; return false; }
                madvise(m, gzmap_n, MADV_SEQUENTIAL);
                gzmap = (const uint8_t *)m;
                sgz = new mcgz::SerialGz(gzmap, gzmap_n);
                if (sgz->start()) ::close(fd);
                else {                                                   // (a header it does not take: zlib's reader decides)
                    delete sgz; sgz = nullptr; munmap(m, gzmap_n); gzmap = nullptr;
                    gz = gzdopen(fd, "rb");
                    if (!gz) { ::close(fd); r_err = std::string("cannot open ") + path; return false; }
                    gzbuffer(gz, 1 << 20);
                }
                compressed = true;
            } else if (has_ext(path, ".gz") || !reg) {                   // (anything that is not a regular file - a pipe - is read through zlib, which passes plain bytes on)
                gz = gzdopen(fd, "rb");
                if (!gz) { ::close(fd); r_err = std::string("cannot open ") + path; return false; }
                gzbuffer(gz, 1 << 20);
                compressed = true;
            } else {
                map_n = (size_t)sb.st_size;
                if (map_n) {
                    void *m = mmap(nullptr, map_n, PROT_READ, MAP_PRIVATE, fd, 0);
                    if (m == MAP_FAILED) { ::close(fd); r_err = std::string("cannot map ") + path; return false; }
                    madvise(m, map_n, MADV_SEQUENTIAL);
                    map = (const uint8_t *)m;
                }
                ::close(fd);
                win = map; len = 0; vend = map + map_n;
                if (t_range_lo >= 0 && map_n) {
                    // A byte window [lo, hi): the records that START in it.  Both ends are moved to the first record start behind them by the
                    // same rule (an '@' line whose second next line starts with '+', or a '>' line), so the windows [b0, b1), [b1, b2), ...
                    // of a file cut it into whole records whoever reads them - the ranks of a multi-GPU run, each with its own sampler.
                    const size_t lo = (size_t)std::min<int64_t>(t_range_lo, (int64_t)map_n), hi = (size_t)std::min<int64_t>(std::max(t_range_hi, t_range_lo), (int64_t)map_n);
                    const int kind = map[0] == '@' ? '@' : map[0] == '>' ? '>' : 0;
                    const size_t s0 = lo == 0 ? 0 : guess_start(map, lo, map_n, 1 << 30, kind), s1 = hi >= map_n ? map_n : guess_start(map, hi, map_n, 1 << 30, kind);
                    win = map + s0; vend = map + std::max(s0, s1); ranged = true;
                }
                if (vend == win) at_end = true;
            }
        }
        if (t_range_lo >= 0 && !ranged) { r_err = std::string("a byte window needs a plain regular file: ") + path; close(); return false; }
        if (t_bz_b0 >= 0) { r_err = std::string("a block range needs a .bz2 file: ") + path; close(); return false; }
        if (compressed && !pgz) {
            if (t_peek) peek_blocks = 8;
            for (auto &r : ring) r.resize(BLK);
            th = std::thread([this] { produce(); });
        }
        // (the parallel inflate has its own workers running ahead of the reader: its bytes are taken straight into the window - extend() -
        // without a producer thread and a ring in between; round 5: one copy of every byte less, a third of the consumer's time)
        return true;
    }
    void close()
    {
        if (compressed && getenv("MC_READER_TIMING") && (t_prod_read + t_ext_copy) > 0) fprintf(stderr, "reader timing: producer inflate+copy %.3f s, producer waits for ring space %.3f s | consumer waits for data %.3f s, consumer copies %.3f s\n", t_prod_read, t_prod_wait, t_ext_wait, t_ext_copy);
        if (th.joinable()) {
            { std::unique_lock<std::mutex> lk(mu); stop = true; cv.notify_all(); }
            th.join();
        }
        const bool gp_t = gp_on && getenv("MC_READER_TIMING");
        const double c0 = gp_t ? now() : 0;
        if (pgz) gz_part_collect();
        const double c1 = gp_t ? now() : 0;
        if (pgz) { delete pgz; pgz = nullptr; }
        const double c2 = gp_t ? now() : 0;
        if (sgz) { delete sgz; sgz = nullptr; }
        if (gzmap) { munmap((void *)gzmap, gzmap_n); gzmap = nullptr; }
        if (gp_t) fprintf(stderr, "gz part [%zu, %zu): rest of the chunks %.3f s, workers joined and buffers handed on %.3f s, unmapped %.3f s\n", gp_k0, gp_k1, c1 - c0, c2 - c1, now() - c2);
        if (gz) { gzclose(gz); gz = nullptr; }
        if (pbz) { delete pbz; pbz = nullptr; }
        if (bzmap) { munmap((void *)bzmap, bzmap_n); bzmap = nullptr; }
        if (bz) { bz->close(); delete bz; bz = nullptr; }
        if (map && part_text.empty()) munmap((void *)map, map_n);
        map = nullptr;
    }
    // The chunks [k0, k1) of a .gz file as a window of its TEXT (mc_reader_open_gz_part; mc_pgzip.h start_slice): all of them - and one
    // more - are decoded speculatively at once; when the owner of the slice in front has told where it ended and what the 32 KB in front of
    // that are (slice 0 knows), the chunks are stitched, the owner of the next slice is told the same, and the text is read like the byte
    // window of a plain file: the records that START in the text of the slice's own chunks, both ends moved to the first record start
    // behind them by the rule of mc_reader_open_range.
    bool open_gz_part(const char *path)
    {
        GzPartCtx *ctx = t_gz_ctx;
        auto fail_out = [&](const std::string &m) { r_err = m; if (ctx) { std::unique_lock<std::mutex> lk(ctx->mu); ctx->out_fail = true; ctx->out_ready = true; ctx->cv.notify_all(); } return false; };
        if (!ctx || !has_ext(path, ".gz")) return fail_out(std::string("a chunk range needs a .gz file: ") + path);
        struct stat sb;
        const int fd = ::open(path, O_RDONLY);
        if (fd < 0 || fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size < 18) { if (fd >= 0) ::close(fd); return fail_out(std::string("cannot map ") + path); }
        void *m = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        ::close(fd);
        if (m == MAP_FAILED) return fail_out(std::string("cannot map ") + path);
        gzmap = (const uint8_t *)m; gzmap_n = (size_t)sb.st_size;
        const size_t k0 = (size_t)t_gz_k0, k1 = (size_t)t_gz_k1;
        pgz = new mcgz::ParallelGz(gzmap, gzmap_n, std::min(std::max(2, inflate_threads()), 32), (size_t)t_gz_chunk);
        if (!pgz->start_slice(k0, k1)) return fail_out(std::string("not a gzip file the parallel reader takes, or no such chunks: ") + path);
        const size_t nchunks = pgz->nchunks();
        if (k0 > 0) {                                                  // where the slice in front ended (the decoding of this one is under way meanwhile)
            std::unique_lock<std::mutex> lk(ctx->mu);
            ctx->cv.wait(lk, [&] { return ctx->in_ready; });
            if (ctx->in_fail) { lk.unlock(); return fail_out("the slice in front of this one failed"); }
            pgz->set_state(ctx->in);
        }
        mcgz::ParallelGz::SliceState out;
        const bool okst = pgz->end_state(out);
        { std::unique_lock<std::mutex> lk(ctx->mu); ctx->out = out; ctx->out_fail = !okst; ctx->out_ready = true; ctx->cv.notify_all(); }
        if (!okst) { r_err = std::string("damaged deflate data in ") + path; return false; }
        // the text is not collected: it is read like any inflated stream (extend -> gz_part_read), the resolve jobs of the workers and
        // the parser's threads running side by side - from the first record start in the slice's first chunk to the first one in the
        // chunk behind the slice (both by guess_start on that ONE chunk's text: the neighbour computes the same place from the same bytes)
        gp_on = true; gp_k0 = k0; gp_k1 = k1; gp_nchunks = nchunks; gp_kind = t_gz_kind; gp_ctx = ctx; gp_path = path;
        gp_ends_here = out.stop || k1 >= nchunks;                      // the data ends in this slice: its text runs to the end
        compressed = true;
        return true;
    }
    bool gp_on = false, gp_ends_here = false, gp_own_set = false, gp_last = false, gp_done = false, gp_collected = false;
    size_t gp_k0 = 0, gp_k1 = 0, gp_nchunks = 0; int gp_kind = 0; GzPartCtx *gp_ctx = nullptr; std::string gp_path;
    const uint8_t *gp_p = nullptr; size_t gp_at = 0, gp_end = 0;
    bool gz_part_next(bool *bad, std::string *msg)
    {
        const size_t idx = pgz->next_chunk_index();
        if (!gp_own_set && idx >= gp_k1) { gp_ctx->nsegs_own = pgz->segs(); gp_own_set = true; }
        if (gp_last) { gp_done = true; return false; }
        size_t n = 0;
        if (!pgz->next_chunk_view(&gp_p, &n)) {
            gp_done = true;
            if (pgz->slice_failed()) { *bad = true; *msg = pgz->slice_error(); }
            return false;
        }
        gp_at = 0; gp_end = n;
        if (idx == gp_k0 && gp_k0 > 0) {
            gp_at = std::min(guess_start(gp_p, 0, n, 1 << 30, gp_kind, true), n);
            if (gp_at >= n && gp_k0 + 1 < gp_nchunks && !gp_ends_here) { gp_done = true; *bad = true; *msg = "no record start within the first chunk of the range (a record longer than a chunk?)"; return false; }
        }
        if (idx >= gp_k1) {                                              // the chunk behind the slice: up to the first record start in it
            gp_last = true;
            if (!gp_ends_here) {
                const size_t s1 = guess_start(gp_p, 0, n, 1 << 30, gp_kind, true);
                if (s1 >= n && gp_k1 + 1 < gp_nchunks) { gp_done = true; *bad = true; *msg = "no record start within a chunk behind the range (a record longer than a chunk?)"; return false; }
                gp_end = std::min(s1, n);
            }
        }
        return true;
    }
    int gz_part_read(uint8_t *dst, int n, bool *bad, std::string *msg)
    {
        int got = 0;
        *bad = false;
        while (got < n) {
            if (gp_at < gp_end) {
                const size_t k = std::min<size_t>((size_t)(n - got), gp_end - gp_at);
                memcpy(dst + got, gp_p + gp_at, k);
                gp_at += k; got += (int)k;
                continue;
            }
            if (gp_done || !gz_part_next(bad, msg)) break;
        }
        return got;
    }
    // after the sampling (which may have stopped early: -n): the rest of the slice's chunks pass by for their CRCs, and the segments go
    // to where mc_reader_gz_finish finds them
    void gz_part_collect()
    {
        if (!gp_on || gp_collected || !pgz) return;
        gp_collected = true;
        bool bad = false; std::string msg;
        gp_at = gp_end = 0;
        while (!gp_done && gz_part_next(&bad, &msg)) gp_at = gp_end = 0;
        if (!gp_own_set) { gp_ctx->nsegs_own = pgz->segs(); gp_own_set = true; }
        gp_ctx->segs = pgz->seg_list();
    }
    // The blocks [b0, b1) of a .bz2 file as a window of its TEXT (mc_reader_open_bz2_part): they are decoded - with two more blocks behind
    // them - into one buffer, which is then read like the byte window of a plain file: the records that START in the text of the rank's own
    // blocks, both ends moved to the first record start behind them by the rule of mc_reader_open_range (the line the boundary lies in is
    // skipped, then the first '@' line whose second next line starts with '+', or the first '>' line) - so consecutive block ranges cut the
    // file into whole records whoever decodes them.  t_bz_kind: '@' or '>', what a record of this file starts with.
    bool open_bz2_part(const char *path)
    {
        struct stat sb;
        const int fd = ::open(path, O_RDONLY);
        if (fd < 0 || fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size < 14) { if (fd >= 0) ::close(fd); r_err = std::string("cannot map ") + path; return false; }
        void *m = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        ::close(fd);
        if (m == MAP_FAILED) { r_err = std::string("cannot map ") + path; return false; }
        bzmap = (const uint8_t *)m; bzmap_n = (size_t)sb.st_size;
        mcbz::Api api;
        api.init = (int (*)(mcbz::BzStreamT *, int, int))g_bz2.init; api.decompress = (int (*)(mcbz::BzStreamT *))g_bz2.decompress; api.end = (int (*)(mcbz::BzStreamT *))g_bz2.end;
        pbz = new mcbz::ParallelBz2(bzmap, bzmap_n, api, std::min(std::max(2, reader_threads()), 32));
        const size_t b0 = (size_t)t_bz_b0, b1 = (size_t)t_bz_b1, EXTRA = 2;
        if (!pbz->start(false) || pbz->tail_byte != bzmap_n || b1 > pbz->blocks.size() || b0 >= b1) { r_err = std::string("not a well-formed .bz2 file, or no such blocks: ") + path; return false; }
        const size_t nb = pbz->blocks.size(), bx = std::min(nb, b1 + EXTRA);
        pbz->run_part(b0, bx);
        size_t own = 0;
        for (size_t k = b0; k < bx; k++) {
            if (k == b1) own = part_text.size();
            if (!pbz->read_block(part_text)) { r_err = std::string("a block of the .bz2 file does not decode: ") + path; return false; }
        }
        if (bx == b1) own = part_text.size();
        delete pbz; pbz = nullptr;
        map_n = part_text.size();
        if (part_text.empty()) part_text.push_back(0);                 // (owned, even when empty: close() must not unmap it)
        map = part_text.data();
        const int kind = t_bz_kind;
        const size_t s0 = b0 == 0 ? 0 : guess_start(map, 0, map_n, 1 << 30, kind, true);
        size_t s1 = map_n;
        if (b1 < nb) {
            s1 = guess_start(map, own, map_n, 1 << 30, kind, true);
            if (s1 >= map_n && bx < nb) { r_err = std::string("no record start within two blocks behind the range (a record longer than a block?): ") + path; return false; }
        }
        win = map + std::min(s0, map_n); vend = map + std::max(std::min(s0, map_n), s1); len = 0; ranged = true;
        if (vend == win) at_end = true;
        return true;
    }
    // .bz2: the blocks of the well-formed streams from the parallel decoder, then - from the first stream that is not, or from the start of
    // a stream one of whose blocks libbz2 refused - the one-stream decoder with its rules (mc_pbzip2.h)
    int bz_read(uint8_t *dst, int want, bool *bad, std::string *msg)
    {
        *bad = false;
        int got = 0;
        if (pbz && !pbz->handover) got = pbz->read(dst, want);
        if (got < want && pbz && pbz->handover && !bz_tail) {
            bz_tail = true;
            bz = new Bz2File();
            if (!bz->open_at(bz_path.c_str(), pbz->handover_byte, pbz->handover_got_any)) { *bad = true; *msg = "read error"; return got; }
            bz_skip = pbz->handover_skip;
        }
        if (got < want && bz) {
            std::vector<uint8_t> scratch;
            while (bz_skip > 0) {                                        // (what the parallel decoder had delivered of this stream already)
                if (scratch.empty()) scratch.resize(1 << 20);
                const int k = (int)std::min<uint64_t>(bz_skip, scratch.size());
                const int r = bz->read(scratch.data(), k, bad, msg);
                bz_skip -= (uint64_t)r;
                if (r < k) return got;                                   // the stream ends (or fails) inside what was delivered: nothing more to give
            }
            got += bz->read(dst + got, want - got, bad, msg);
        }
        return got;
    }
    void produce()
    {
        for (;;) {
            int slot;
            {
                const double w0 = now();
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [this] { return count < NBLK || stop; });
                t_prod_wait += now() - w0;
                if (stop) return;
                slot = tail;
            }
            const double r0 = now();
            int n;
            bool bad = false;
            std::string msg;
            const int want = peek_blocks > 0 ? (int)BLK_PEEK : (int)BLK;     // (a reader that may stop after a few records gets its first bytes early)
            if (peek_blocks > 0) peek_blocks--;
            if (pgz) {
                n = pgz->read(ring[slot].data(), want, &bad, &msg);
            } else if (sgz) {
                n = sgz->read(ring[slot].data(), want, &bad, &msg);
            } else if (gz) {
                n = gzread(gz, ring[slot].data(), (unsigned)want);
                if (n < want) {
                    // gzip.open raises EOFError / BadGzipFile on a truncated or corrupt stream (reference :47-59); zlib reports a
                    // truncated stream as Z_BUF_ERROR, a damaged one as Z_DATA_ERROR - a clean end leaves Z_OK / Z_STREAM_END and gzeof
                    int errnum = 0;
                    const char *m = gzerror(gz, &errnum);
                    if (n < 0 || (errnum != Z_OK && errnum != Z_STREAM_END) || !gzeof(gz)) { bad = true; msg = m ? m : "read error"; }
                }
            } else {
                n = bz_read(ring[slot].data(), want, &bad, &msg);
            }
            t_prod_read += now() - r0;
            std::unique_lock<std::mutex> lk(mu);
            if (n > 0) { ring_n[slot] = (size_t)n; tail = (tail + 1) % NBLK; count++; }
            if (n < want) { prod_done = true; prod_err = bad; prod_msg = msg; cv.notify_all(); return; }
            cv.notify_all();
        }
    }
    // make the window at least `want` bytes long if the stream has them
    void extend(size_t want)
    {
        if (!compressed) {
            const size_t have = (size_t)(vend - win);
            len = std::min(want, have);
            if (len == have) at_end = true;
            return;
        }
        while (pgz && len < want && !at_end) {
            const double c0 = now();
            const size_t n = BLK;
            if (buf_at + len + n > buf.size()) {                         // make room: move the window to the front, grow if needed
                if (buf_at) { memmove(buf.data(), buf.data() + buf_at, len); buf_at = 0; }
                if (len + n > buf.size()) buf.resize(std::max(len + n, buf.size() * 2));
            }
            bool bad = false;
            std::string msg;
            const int got = gp_on ? gz_part_read(buf.data() + buf_at + len, (int)n, &bad, &msg) : pgz->read(buf.data() + buf_at + len, (int)n, &bad, &msg);
            if (got > 0) len += (size_t)got;
            if (got < (int)n) { at_end = true; failed = bad; prod_err = bad; prod_msg = msg; }
            t_ext_copy += now() - c0;
        }
        while (!pgz && len < want && !at_end) {
            const double w0 = now();
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [this] { return count > 0 || prod_done; });
            t_ext_wait += now() - w0;
            if (count == 0) { at_end = true; failed = prod_err; break; }
            const size_t n = ring_n[head];
            lk.unlock();
            const double c0 = now();
            if (buf_at + len + n > buf.size()) {                         // make room: move the window to the front, grow if needed
                if (buf_at) { memmove(buf.data(), buf.data() + buf_at, len); buf_at = 0; }
                if (len + n > buf.size()) buf.resize(std::max(len + n, buf.size() * 2));
            }
            memcpy(buf.data() + buf_at + len, ring[head].data(), n);
            len += n;
            t_ext_copy += now() - c0;
            lk.lock();
            head = (head + 1) % NBLK; count--;
            cv.notify_all();
        }
        win = buf.data() + buf_at;
    }
    void consume(size_t k)
    {
        if (!compressed) { win += k; len -= k; return; }
        buf_at += k; len -= k; win = buf.data() + buf_at;
    }
};

// [p, p + n) cut back to whole lines (0: no line end found)
size_t whole_lines(const uint8_t *p, size_t n)
{
    for (size_t i = n; i > 0; i--) {
        if (p[i - 1] == '\n') return i;
        if (n - i > ((size_t)16 << 20)) break;
    }
    for (size_t i = n; i > 1; i--) if (p[i - 2] == '\r' && p[i - 1] != '\n') return i - 1;   // lone-CR files: a '\r' whose successor is known
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// parse_seqs as a restartable state machine over a region of whole lines, and the per-record evaluation
// ------------------------------------------------------------------------------------------------------------------
// R_PASS / R_LOWQ / R_DUP / R_ERR: the sampler's verdict, given off the file-order walk - by the parser's threads (Params::decide: no -d,
// a record's fate depends on nothing but itself) or by the walkers of the duplicate classes (Params::shards: -d).  R_ERR: the reference
// raises at this record if its sampler gets that far (err_text)
enum : uint8_t { R_SHORT = 1, R_QUAL = 2, R_RCBAD = 4, R_PASS = 8, R_LOWQ = 16, R_DUP = 32, R_ERR = 64,
                 R_STABLE = 128 };                      // the sequence lies in a plain file's mapping, which outlives the sampler's set (no copy kept)
enum { NSHARD = 64 };                                  // duplicate classes are dealt to this many sets by the top bits of their key (Params::shards)
struct Rec {
    const uint8_t *seq; const uint8_t *qual;
    uint32_t len, qlen;
    uint32_t ncount, nq; int32_t qmin; int64_t qsum;   // first L bases / first min(L, qlen) qualities
    uint64_t h1, h2;                                   // hashes of the sequence and of its reverse complement (only with -d)
    uint32_t out;                                      // slot in the output of this region (or ~0)
    uint8_t flags;
    uint8_t shard;                                     // of the record's duplicate class: top bits of min(h1, h2) - a sequence and its reverse complement share it
};

struct ShItem { uint64_t h1, h2; const uint8_t *seq; uint32_t len, idx; uint8_t q; };   // q: in - the quality filter's verdict (R_PASS / R_LOWQ / R_ERR) | R_STABLE; out - the sampler's (R_DUP or that verdict)

struct Arena {   // stable storage for records whose sequence spans several lines
    std::vector<std::unique_ptr<uint8_t[]>> blocks; size_t at = 0, cap = 0;
    uint8_t *alloc(size_t n)
    {
        if (at + n > cap) { cap = std::max<size_t>(n, 1 << 20); blocks.emplace_back(new uint8_t[cap]); at = 0; }
        uint8_t *p = blocks.back().get() + at; at += n; return p;
    }
    void clear() { blocks.clear(); at = cap = 0; }
};

struct Piece {
    size_t start = 0, stop = 0, end = 0;               // offsets into the region: first byte, guessed start of the next piece, first unconsumed byte
    bool done = false;                                  // the parser reached its terminal state (the generator returned)
    bool ragged = false;                                // ... in the middle of a record (the data ended inside its sequence or qualities)
    std::vector<Rec> recs;
    Arena arena;
    int64_t bases = 0;
    // with Params::decide / shards: how many of the piece's records are too short / fail the quality filter / pass / are duplicates, and
    // whether one of them is a record the sampler would raise an error at (then the piece is walked record by record, in file order)
    int64_t n_short = 0, n_lowq = 0, n_pass = 0, n_dup = 0;
    bool anomaly = false;
    // with Params::shards: the piece's long-enough records grouped by the shard of their duplicate class, file order inside a shard -
    // everything a class's walker needs of a record in one place, read in a stream (the records themselves lie in another core's cache)
    std::vector<ShItem> sh_items; uint32_t sh_off[NSHARD + 1] = {};
    int qoff_answer = 0;                               // with Params::find_qoff: 32 / 64 by the piece's first quality character that decides, -2 a record without qualities came first, 0 nothing decides
    // for the next region: the vectors keep their memory (fresh ones are page faults - several times the parse itself in a process's first run)
    void reset()
    {
        start = stop = end = 0; done = ragged = anomaly = false; recs.clear(); sh_items.clear(); if (!arena.blocks.empty()) arena.clear(); qoff_answer = 0;
        bases = n_short = n_lowq = n_pass = n_dup = 0;
    }
};

struct Line { const uint8_t *p; size_t n; bool nl; size_t off; };
// next line of [pos, e): content without terminator; returns false at e
// first '\n' or '\r' of [b, b + n), or null: ONE pass over the line, 16 bytes at a time (two memchr calls per line - the lines of a FASTQ
// file are 150 bytes, the calls cost more than the bytes - were half of the parser's time)
inline const uint8_t *scan_eol(const uint8_t *b, size_t n)
{
#if defined(__SSE2__)
    const __m128i nl = _mm_set1_epi8('\n'), cr = _mm_set1_epi8('\r');
    size_t i = 0;
    for (; i + 16 <= n; i += 16) {
        const __m128i v = _mm_loadu_si128((const __m128i *)(b + i));
        const int m = _mm_movemask_epi8(_mm_or_si128(_mm_cmpeq_epi8(v, nl), _mm_cmpeq_epi8(v, cr)));
        if (m) return b + i + __builtin_ctz((unsigned)m);
    }
    for (; i < n; i++) if (b[i] == '\n' || b[i] == '\r') return b + i;
    return nullptr;
#else
    const uint8_t *q = (const uint8_t *)memchr(b, '\n', n);
    const size_t lim = q ? (size_t)(q - b) : n;
    const uint8_t *c = lim ? (const uint8_t *)memchr(b, '\r', lim) : nullptr;
    return c ? c : q;
#endif
}
inline bool next_line(const uint8_t *base, size_t &pos, size_t e, Line &ln)
{
    if (pos >= e) return false;
    const uint8_t *b = base + pos;
    const size_t avail = e - pos;
    const uint8_t *q = scan_eol(b, avail);
    ln.p = b; ln.off = pos;
    if (!q) { ln.n = avail; ln.nl = false; pos = e; return true; }
    ln.n = (size_t)(q - b); ln.nl = true;
    pos += ln.n + 1;
    if (*q == '\r' && pos < e && base[pos] == '\n') pos++;
    return true;
}

// 64-bit hash of a byte string, and the same hash of its reverse complement read off the string backwards (both only with -d).
// Four lanes - word i of the string goes to lane i mod 4 - so that four multiplications are in flight instead of one chain of n / 8
// dependent ones (a 300-base sequence: 38 words; the two hashes were a third of the parser's time with -d).
#define MC_HMIX(x, w) do { (x) = ((x) ^ (w)) * 0x9FB21C651E98DF25ull; (x) ^= (x) >> 29; } while (0)
struct H4 {
    uint64_t a, b, c, d;
    explicit H4(size_t n)
    {
        const uint64_t s = 0x9E3779B97F4A7C15ull ^ (n * 0xff51afd7ed558ccdull);
        a = s; b = s ^ 0xD6E8FEB86659FD93ull; c = s ^ 0xA0761D6478BD642Full; d = s ^ 0xE7037ED1A0B428DBull;
    }
    void lane(unsigned k, uint64_t w) { switch (k & 3) { case 0: MC_HMIX(a, w); break; case 1: MC_HMIX(b, w); break; case 2: MC_HMIX(c, w); break; default: MC_HMIX(d, w); } }
    uint64_t fin() const
    {
        uint64_t h = a;
        h = (h ^ (b << 21 | b >> 43)) * 0x9FB21C651E98DF25ull; h ^= h >> 29;
        h = (h ^ (c << 42 | c >> 22)) * 0x9FB21C651E98DF25ull; h ^= h >> 29;
        h = (h ^ (d << 11 | d >> 53)) * 0x9FB21C651E98DF25ull; h ^= h >> 29;
        h *= 0xc4ceb9fe1a85ec53ull; h ^= h >> 32;
        return h;
    }
};
uint64_t h64(const uint8_t *p, size_t n)
{
    H4 h(n);
    size_t i = 0;
    for (; i + 32 <= n; i += 32) {
        uint64_t w0, w1, w2, w3;
        memcpy(&w0, p + i, 8); memcpy(&w1, p + i + 8, 8); memcpy(&w2, p + i + 16, 8); memcpy(&w3, p + i + 24, 8);
        MC_HMIX(h.a, w0); MC_HMIX(h.b, w1); MC_HMIX(h.c, w2); MC_HMIX(h.d, w3);
    }
    unsigned k = 0;
    for (; i + 8 <= n; i += 8, k++) { uint64_t w; memcpy(&w, p + i, 8); h.lane(k, w); }
    if (i < n) { uint64_t w = 0; memcpy(&w, p + i, n - i); h.lane(k, w); }
    return h.fin();
}
struct RcTab { uint8_t t[256]; RcTab() { memset(t, 0, sizeof t); t['A'] = 'T'; t['T'] = 'A'; t['G'] = 'C'; t['C'] = 'G'; t['N'] = 'N'; } };
const RcTab g_rc;
// h64 of the reverse complement, read off the sequence backwards; false: a base outside ACGTN (the value is of no use then).
// 16 bases at a time where SSE2 is there: complement = the byte XOR 0x15 for A / T, XOR 0x04 for C / G, N as it is; a word of the
// reverse complement is the byte-swapped word of the complement.
bool h64_rc(const uint8_t *p, size_t n, uint64_t *out)
{
    H4 h(n);
    uint8_t all = 0xFF;
    size_t i = 0;
    unsigned k = 0;                                                  // the next word's lane
    const uint8_t *q = p + n;
#if defined(__SSE2__)
    {
        const __m128i cA = _mm_set1_epi8('A'), cC = _mm_set1_epi8('C'), cG = _mm_set1_epi8('G'), cT = _mm_set1_epi8('T'), cN = _mm_set1_epi8('N');
        const __m128i xAT = _mm_set1_epi8(0x15), xCG = _mm_set1_epi8(0x04);
        int ok = 0xFFFF;
        auto comp16 = [&](const uint8_t *src, uint64_t &first, uint64_t &second) {   // the two words of the reverse complement that 16 bases make
            const __m128i v = _mm_loadu_si128((const __m128i *)src);
            const __m128i at = _mm_or_si128(_mm_cmpeq_epi8(v, cA), _mm_cmpeq_epi8(v, cT)), cg = _mm_or_si128(_mm_cmpeq_epi8(v, cC), _mm_cmpeq_epi8(v, cG));
            ok &= _mm_movemask_epi8(_mm_or_si128(_mm_or_si128(at, cg), _mm_cmpeq_epi8(v, cN)));
            const __m128i c = _mm_xor_si128(v, _mm_or_si128(_mm_and_si128(at, xAT), _mm_and_si128(cg, xCG)));
            first = __builtin_bswap64((uint64_t)_mm_cvtsi128_si64(_mm_unpackhi_epi64(c, c))); second = __builtin_bswap64((uint64_t)_mm_cvtsi128_si64(c));
        };
        for (; i + 32 <= n; i += 32) {
            q -= 32;
            uint64_t w0, w1, w2, w3;
            comp16(q + 16, w0, w1); comp16(q, w2, w3);
            MC_HMIX(h.a, w0); MC_HMIX(h.b, w1); MC_HMIX(h.c, w2); MC_HMIX(h.d, w3);
        }
        if (i + 16 <= n) { q -= 16; uint64_t w0, w1; comp16(q, w0, w1); h.lane(0, w0); h.lane(1, w1); i += 16; k = 2; }
        if (ok != 0xFFFF) all = 0;
    }
#endif
    for (; i + 8 <= n; i += 8, k++) {
        uint64_t w = 0;
        for (int j = 0; j < 8; j++) { const uint8_t d = g_rc.t[*--q]; all &= (uint8_t)(d ? 0xFF : 0); w |= (uint64_t)d << (8 * j); }
        h.lane(k, w);
    }
    if (i < n) {
        uint64_t w = 0;
        for (int j = 0; i < n; i++, j++) { const uint8_t d = g_rc.t[*--q]; all &= (uint8_t)(d ? 0xFF : 0); w |= (uint64_t)d << (8 * j); }
        h.lane(k, w);
    }
    *out = h.fin();
    return all != 0 || n == 0;
}

struct Params { size_t L = 0; int fastq = 0, qoff = 0, dups = 0; bool count_only = false;
                bool decide = false; double max_unknown = 0, mean_q = 0, min_q = 0;   // decide: the quality filter's verdict per record is given by the parser's threads (no -d: a record's fate depends on nothing but itself)
                bool shards = false;
                bool stable = false;
                bool find_qoff = false; };                                            // find_qoff: mc_quality_offset - every piece looks for its first quality character that decides                                               // stable: the file is a mapping that stays until the sampler ends                                               // shards: -d - the parser's threads group their records by duplicate class, the classes' walkers give the verdicts

void evaluate(Rec &r, const Params &P)
{
    r.flags = (r.qual ? R_QUAL : 0); r.out = ~0u; r.shard = 0;
    r.ncount = 0; r.nq = 0; r.qmin = 1 << 30; r.qsum = 0; r.h1 = r.h2 = 0;
    if (P.count_only) return;
    if (r.len < P.L) { r.flags |= R_SHORT; return; }
    // the first L bases: how many are 'N'; the first min(L, qlen) qualities: their sum and their minimum (as ord(c) - offset) - 16 bytes at
    // a time where SSE2 is there (byte sums by psadbw, the minimum by pminub: the offset comes off at the end), the plain loops elsewhere
    uint32_t nc = 0;
    size_t i = 0;
#if defined(__SSE2__)
    {
        const __m128i cN = _mm_set1_epi8('N'), zero = _mm_setzero_si128();
        __m128i acc = zero;
        for (; i + 16 <= P.L; i += 16)
            acc = _mm_add_epi64(acc, _mm_sad_epu8(_mm_and_si128(_mm_cmpeq_epi8(_mm_loadu_si128((const __m128i *)(r.seq + i)), cN), _mm_set1_epi8(1)), zero));
        nc = (uint32_t)(_mm_cvtsi128_si64(acc) + _mm_cvtsi128_si64(_mm_unpackhi_epi64(acc, acc)));
    }
#endif
    for (; i < P.L; i++) nc += (r.seq[i] == 'N');
    r.ncount = nc;
    if (P.fastq && r.qual) {
        const size_t nq = r.qlen < P.L ? r.qlen : P.L;
        int64_t sum = 0; int mn = 1 << 30;
        size_t k = 0;
#if defined(__SSE2__)
        if (nq >= 16) {
            const __m128i zero = _mm_setzero_si128();
            __m128i acc = zero, vmin = _mm_set1_epi8((char)0xFF);
            for (; k + 16 <= nq; k += 16) {
                const __m128i v = _mm_loadu_si128((const __m128i *)(r.qual + k));
                acc = _mm_add_epi64(acc, _mm_sad_epu8(v, zero));
                vmin = _mm_min_epu8(vmin, v);
            }
            sum = (int64_t)(_mm_cvtsi128_si64(acc) + _mm_cvtsi128_si64(_mm_unpackhi_epi64(acc, acc))) - (int64_t)k * P.qoff;
            vmin = _mm_min_epu8(vmin, _mm_srli_si128(vmin, 8)); vmin = _mm_min_epu8(vmin, _mm_srli_si128(vmin, 4));
            vmin = _mm_min_epu8(vmin, _mm_srli_si128(vmin, 2)); vmin = _mm_min_epu8(vmin, _mm_srli_si128(vmin, 1));
            mn = (int)(_mm_cvtsi128_si32(vmin) & 0xFF) - P.qoff;
        }
#endif
        for (; k < nq; k++) { const int q = (int)r.qual[k] - P.qoff; sum += q; mn = q < mn ? q : mn; }
        r.nq = (uint32_t)nq; r.qsum = sum; r.qmin = mn;
    }
    if (P.dups) { r.h1 = h64(r.seq, r.len); if (!h64_rc(r.seq, r.len, &r.h2)) r.flags |= R_RCBAD; r.shard = (uint8_t)((r.h1 < r.h2 ? r.h1 : r.h2) >> 58); }
}
// quality_filter (reference :269-291) as mc_reader_run's loop states it, for one record that is long enough; false: the sampler
// would raise at this record
inline bool decide(Rec &r, const Params &P)
{
    bool fail = (double)(100 * (long long)r.ncount) / (double)P.L > P.max_unknown;
    if (!fail && P.fastq) {
        if (!(r.flags & R_QUAL) || r.nq == 0) return false;
        if ((double)r.qsum / (double)r.nq < P.mean_q) fail = true;
        else if ((double)r.qmin < P.min_q) fail = true;
    }
    r.flags |= fail ? R_LOWQ : R_PASS;
    return true;
}

// Runs parse_seqs (reference :294-325) over region[start, e) until a header line at an offset >= stop would be consumed (or
// the region ends).  eof: the region ends the file (else an unfinished record at its end is left for the next region:
// pc.end = offset of its header line).
void parse_piece_lines(const uint8_t *base, size_t e, bool eof, Piece &pc, const Params &P);
void parse_piece(const uint8_t *base, size_t e, bool eof, Piece &pc, const Params &P)
{
    parse_piece_lines(base, e, eof, pc, P);
    if (P.find_qoff) {                                    // auto_detect_quality_offset (:175-187), the piece's share: its records in order, their qualities character by character
        for (const Rec &r : pc.recs) {
            if (!r.qual) { pc.qoff_answer = -2; break; }
            for (uint32_t i = 0; i < r.qlen; i++) {
                const uint8_t ch = r.qual[i];
                if (ch >= '!' && ch <= '9') { pc.qoff_answer = 32; break; }
                if (ch >= 'K' && ch <= '~') { pc.qoff_answer = 64; break; }
            }
            if (pc.qoff_answer) break;
        }
        return;
    }
    if (!P.shards) return;
    // -d: the piece's long-enough records grouped by the shard of their duplicate class (a counting sort; file order inside a shard),
    // each with the quality filter's verdict - which counts if the record turns out to be nobody's duplicate.
    // A record with a base outside ACGTN can be nobody's duplicate - a sequence enters the set only behind its own reverse_complement(),
    // which raises for such a base - so it has its verdict already: the reference raises there
    uint32_t cnt[NSHARD] = {};
    for (Rec &r : pc.recs) {
        if (r.flags & R_SHORT) continue;
        if (r.flags & R_RCBAD) { r.flags |= R_ERR; continue; }
        cnt[r.shard]++;
    }
    uint32_t at = 0;
    for (int s = 0; s < NSHARD; s++) { pc.sh_off[s] = at; at += cnt[s]; cnt[s] = pc.sh_off[s]; }
    pc.sh_off[NSHARD] = at;
    pc.sh_items.resize(at);
    for (size_t i = 0; i < pc.recs.size(); i++) {
        Rec &r = pc.recs[i];
        if (r.flags & (R_SHORT | R_RCBAD)) continue;
        Rec t = r;
        const uint8_t q = !decide(t, P) ? R_ERR : (t.flags & (R_PASS | R_LOWQ));
        pc.sh_items[cnt[r.shard]++] = ShItem{r.h1, r.h2, r.seq, r.len, (uint32_t)i, (uint8_t)(q | (r.flags & R_STABLE))};
    }
}
void parse_piece_lines(const uint8_t *base, size_t e, bool eof, Piece &pc, const Params &P)
{
    size_t pos = pc.start;
    const size_t stop = pc.stop;
    Line ln;
    bool have_hdr = false;
    size_t hdr_off = 0;
    auto chomped = [](const Line &l) { return (!l.nl && l.n > 0) ? l.n - 1 : l.n; };   // len(line[:-1])
    auto emit = [&](const uint8_t *s, size_t sn, const uint8_t *q, size_t qn, bool hasq) {
        Rec r; r.seq = s; r.len = (uint32_t)sn; r.qual = hasq ? q : nullptr; r.qlen = hasq ? (uint32_t)qn : 0;
        evaluate(r, P);
        if (P.stable && s >= base && s < base + e) r.flags |= R_STABLE;   // (not a sequence joined from several lines in the piece's arena)
        if (P.decide) {
            if (r.flags & R_SHORT) pc.n_short++;
            else if (!decide(r, P)) { r.flags |= R_ERR; pc.anomaly = true; }
            else if (r.flags & R_PASS) pc.n_pass++;
            else pc.n_lowq++;
        }
        pc.bases += (int64_t)sn;
        pc.recs.push_back(r);
    };
    for (;;) {
        if (!have_hdr) {                                             // search for the start of the next record
            bool found = false;
            while (next_line(base, pos, e, ln)) {
                if (ln.n > 0 && (ln.p[0] == '>' || ln.p[0] == '@')) {
                    if (ln.off >= stop) { pc.end = ln.off; return; }
                    found = true; hdr_off = ln.off;
                    break;
                }
            }
            if (!found) { pc.end = e; if (eof) pc.done = true; return; }
            if (chomped(ln) == 0) { pc.end = e; pc.done = true; return; }   // a lone '>' / '@' without newline ends the file: `if not last: break`
        }
        have_hdr = false;
        // the sequence: lines up to one that starts with '@', '+' or '>'
        const uint8_t *s0 = nullptr; size_t sn = 0; int nparts = 0; uint8_t *joined = nullptr; size_t jcap = 0;
        bool got_next = false;
        Line nx{};
        for (;;) {
            if (!next_line(base, pos, e, ln)) break;
            if (ln.n > 0 && (ln.p[0] == '@' || ln.p[0] == '+' || ln.p[0] == '>')) { nx = ln; got_next = true; break; }
            const size_t m = chomped(ln);
            if (nparts == 0) { s0 = ln.p; sn = m; }
            else {                                                   // a sequence over several lines is joined in the piece's arena
                if (nparts == 1) { jcap = std::max<size_t>((sn + m) * 2, 256); joined = pc.arena.alloc(jcap); if (sn) memcpy(joined, s0, sn); }
                else if (sn + m > jcap) { const size_t nc = (sn + m) * 2; uint8_t *nj = pc.arena.alloc(nc); memcpy(nj, joined, sn); joined = nj; jcap = nc; }
                memcpy(joined + sn, ln.p, m); sn += m;
            }
            nparts++;
        }
        const uint8_t *seq = nparts <= 1 ? s0 : joined;
        if (!got_next) {                                              // ran out of lines
            if (!eof) { pc.end = hdr_off; return; }                  // unfinished: the next region starts again at its header
            emit(seq, sn, nullptr, 0, false); pc.end = e; pc.done = true; pc.ragged = P.fastq != 0; return;
        }
        const size_t nxn = chomped(nx);
        if (nxn == 0) { emit(seq, sn, nullptr, 0, false); pc.end = e; pc.done = true; return; }   // '' : record goes out, the generator returns
        if (nx.p[0] != '+') {                                          // a FASTA record; nx is the next header
            emit(seq, sn, nullptr, 0, false);
            if (nx.off >= stop) { pc.end = nx.off; return; }
            have_hdr = true; hdr_off = nx.off;
            continue;
        }
        // qualities: lines until they cover the sequence
        const uint8_t *q0 = nullptr; size_t qn = 0; int qparts = 0; uint8_t *qj = nullptr; size_t qcap = 0;
        bool complete = false;
        for (;;) {
            if (!next_line(base, pos, e, ln)) break;
            const size_t m = chomped(ln);
            if (qparts == 0) { q0 = ln.p; qn = m; }
            else {
                if (qparts == 1) { qcap = std::max<size_t>((qn + m) * 2, 256); qj = pc.arena.alloc(qcap); if (qn) memcpy(qj, q0, qn); }
                else if (qn + m > qcap) { const size_t nc = (qn + m) * 2; uint8_t *nj = pc.arena.alloc(nc); memcpy(nj, qj, qn); qj = nj; qcap = nc; }
                memcpy(qj + qn, ln.p, m); qn += m;
            }
            qparts++;
            if (qn >= sn) { complete = true; break; }
        }
        if (complete) { emit(seq, sn, qparts <= 1 ? q0 : qj, qn, true); continue; }
        if (!eof) { pc.end = hdr_off; return; }
        emit(seq, sn, nullptr, 0, false); pc.end = e; pc.done = true; pc.ragged = true; return;   // the file ends inside the qualities
    }
}

// guessed record start at or behind `from`: offset of a '@' line whose second next line starts with '+', or of a '>' line
// kind: 0 either; '>' / '@': only that kind of record start - the end of a byte window (mc_reader_open_range) must not be taken for
// a quality line that happens to start with '>' in a FASTQ file (a piece's guess is verified when the pieces are stitched, a
// window's only by how its parse ends), so there the first byte of the file decides
size_t guess_start(const uint8_t *base, size_t from, size_t e, int max_lines = 64, int kind = 0, bool skip_first)
{
    size_t pos = from;
    Line ln;
    if (pos > 0 || skip_first) { if (!next_line(base, pos, e, ln)) return e; }     // (skip the line `from` points into; skip_first: also when the buffer begins there - the text of a .bz2 block range)
    for (int tries = 0; tries < max_lines; tries++) {
        const size_t at = pos;
        if (!next_line(base, pos, e, ln)) return e;
        if (ln.n == 0) continue;
        if (ln.p[0] == '>' && kind != '@') return at;
        if (ln.p[0] == '@' && kind != '>') {
            size_t p2 = pos; Line a, b;
            if (next_line(base, p2, e, a) && next_line(base, p2, e, b) && b.n > 0 && b.p[0] == '+' && !(a.n > 0 && (a.p[0] == '@' || a.p[0] == '>' || a.p[0] == '+'))) return at;
        }
    }
    return e;
}

// Anonymous memory by the megabyte, straight from the kernel (null on failure).  (Asked to be backed by huge pages - MADV_HUGEPAGE, one
// fault per 2 MB - it was SLOWER here: with transparent huge pages' defrag on "madvise" every such fault may compact memory first;
// sampler + copies of 600 k records 0.04 -> 0.10 - 1.3 s.)
struct BigPages {
    static constexpr size_t HP = (size_t)2 << 20;
    static uint8_t *get(size_t &bytes)
    {
        bytes = (bytes + HP - 1) & ~(HP - 1);
        uint8_t *p = (uint8_t *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        return p == (uint8_t *)MAP_FAILED ? nullptr : p;
    }
};

// Exact set of the accepted (untrimmed) sequences for -d: the strings live back to back in blocks that never move, an
// open-addressing table holds (hash, pointer) in one 16-byte slot - no allocation per sequence, one cache miss per lookup (which
// the walkers of the classes ask for ahead of time: prefetch); equality is decided on the bytes.
struct SeqSet {
    struct Slot { uint32_t h, len; const uint8_t *p; };             // h: the hash's lower half (its low bits place the slot: tables of up to 2^32 slots); p null = empty slot
    std::vector<Slot> tab;
    std::vector<std::pair<uint8_t *, size_t>> blocks; size_t at = 0, cap = 0;
    size_t used = 0;
    SeqSet() = default;
    SeqSet(const SeqSet &) = delete;
    SeqSet &operator=(const SeqSet &) = delete;
    ~SeqSet() { for (auto &b : blocks) munmap(b.first, b.second); }
    void grow()
    {
        const size_t n = tab.empty() ? (1u << 12) : tab.size() * 2;
        std::vector<Slot> nt(n, Slot{0, 0, nullptr});
        for (const Slot &sl : tab) if (sl.p) { size_t j = sl.h & (n - 1); while (nt[j].p) j = (j + 1) & (n - 1); nt[j] = sl; }
        tab.swap(nt);
    }
    void prefetch(uint64_t h) const { if (!tab.empty()) __builtin_prefetch(&tab[h & (tab.size() - 1)]); }
    // is s (rc = false) or the reverse complement of s (rc = true) in the set?  h = the matching hash
    bool contains(uint64_t h, const uint8_t *s, size_t n, bool rc) const
    {
        if (tab.empty()) return false;
        const uint32_t hh = (uint32_t)h;
        for (size_t j = h & (tab.size() - 1);; j = (j + 1) & (tab.size() - 1)) {
            const uint8_t *q = tab[j].p;
            if (!q) return false;
            if (tab[j].h == hh && tab[j].len == n) {
                if (!rc) { if (memcmp(q, s, n) == 0) return true; }
                else { size_t i = 0; for (; i < n; i++) if (q[i] != g_rc.t[s[n - 1 - i]]) break; if (i == n) return true; }
            }
        }
    }
    // (the caller has checked that it is not there); stable: the bytes outlive the set - no copy; false: out of memory
    bool insert(uint64_t h, const uint8_t *s, size_t n, bool stable)
    {
        if ((used + 1) * 2 > tab.size()) { try { grow(); } catch (const std::bad_alloc &) { return false; } }
        const uint8_t *o = s;
        if (!stable) {
            const size_t need = (n + 7) & ~(size_t)7;
            if (at + need > cap) {
                size_t want = std::max<size_t>(need, BigPages::HP);
                uint8_t *b = BigPages::get(want);
                if (!b) return false;
                blocks.emplace_back(b, want); cap = want; at = 0;
            }
            uint8_t *w = blocks.back().first + at; at += need;
            memcpy(w, s, n);
            o = w;
        }
        size_t j = h & (tab.size() - 1);
        while (tab[j].p) j = (j + 1) & (tab.size() - 1);
        tab[j] = Slot{(uint32_t)h, (uint32_t)n, o}; used++;
        return true;
    }
};

// Worker threads of the reader: MC_READER_THREADS in the environment, else the caller's cap (mc_set_host_threads: run_pipeline
// passes args['threads'], the reference's -t), else the CPUs the process may use (cgroup quota), up to 32.
std::atomic<int> g_host_threads{0};
// CPUs this process may actually use: the machine's, or the container's CPU quota when that is smaller (cgroup cpu.max / cfs quota:
// a box that shows 256 CPUs may be allowed 16 of them - threads beyond the quota only get throttled)
int effective_cores()
{
    static const int cached = [] {
        const unsigned hc = std::thread::hardware_concurrency();
        int n = hc ? (int)hc : 1;
        long long quota = -1, period = 0;
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[64] = {0};
            if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
            fclose(f);
        } else {
            if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &quota) != 1) quota = -1; fclose(g); }
            if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &period) != 1) period = 0; fclose(g); }
        }
        if (quota > 0 && period > 0) { const int c = (int)((quota + period - 1) / period); if (c >= 1 && c < n) n = c; }
        return n;
    }();
    return cached;
}
int reader_threads()
{
    if (const char *e = getenv("MC_READER_THREADS")) { const int v = atoi(e); if (v >= 1) return v > 256 ? 256 : v; }
    const int cap = g_host_threads.load();
    if (cap >= 1) return cap > 256 ? 256 : cap;
    // ... but no more than the CPUs the process may USE (a box that shows 256 CPUs and grants 16 - cgroup quota): the quota throttles every
    // thread of the process once it is spent, the one that drives the GPU too - with 32 parser threads on 16 granted CPUs the gapped stage of
    // a streamed range took 63 - 108 ms instead of 47.5 (its launches wait for the host between them) and file -> AGS of a plain FASTQ ran at
    // 48 - 54 M reads/s instead of 55 - 62 with 16 (round 6, tools/e2e_probe.py)
    return std::max(1, std::min(effective_cores(), 32));
}
// inflate workers of a .gz input: they keep their cores busy all the time (the parser's threads mostly wait), so no more of them
// than three quarters of the CPUs the process may use
int inflate_threads()
{
    if (const char *v = getenv("MC_READER_GZ_THREADS")) { const int k = atoi(v); if (k >= 1) return k > 64 ? 64 : k; }
    const int t = reader_threads();
    return std::max(1, std::min(t, std::max(2, effective_cores() * 3 / 4)));
}

// One file, region by region.  on_region(pieces) sees the stitched pieces of a region in file order and returns false to stop
// the file early (sampler full).  Returns 0, or -1 (I/O) / -3 (the reference would have raised).
// seconds of the sampler's own thread by phase (mc_reader_times): waiting for / inflating input, guessing the pieces' starts, the
// parse (all workers), stitching, what on_region does (verdicts, places, copies), of it the walkers of the duplicate classes (-d)
struct WalkTimes { double v[6] = {0, 0, 0, 0, 0, 0}; };
typedef std::vector<std::pair<const uint8_t *, size_t>> KeptMaps;
int walk_file(const std::string &path, const Params &P0, Pool &pool, const std::function<bool(std::vector<Piece *> &)> &on_region, KeptMaps *keep = nullptr, WalkTimes *wt = nullptr)
{
    const bool wt_on = wt != nullptr;
    WalkTimes wt_none;
    double *g_wt = wt ? wt->v : wt_none.v;
    auto wnow = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    Stream st;
    if (!st.open(path.c_str())) return st.bad_gzip ? -3 : -1;
    // keep: a plain file's mapping is handed to the caller instead of being unmapped - the sampler's set of accepted sequences (-d) points
    // into it rather than holding copies
    struct Keeper { Stream &st; KeptMaps *keep; ~Keeper() { if (keep && st.map) { keep->emplace_back(st.map, st.map_n); st.map = nullptr; } } } keeper{st, keep};
    Params P = P0;
    P.stable = keep != nullptr && !st.compressed && st.map != nullptr;
    const int T = pool.size();
    size_t region_bytes = (size_t)std::max(1, std::min(T, 16)) * ((size_t)4 << 20), piece_bytes = (size_t)256 << 10;
    const size_t full_region = region_bytes;
    if (t_peek) region_bytes = (size_t)1 << 16;                    // grows to the full size region by region
    if (const char *v = getenv("MC_READER_REGION_BYTES")) region_bytes = std::max<size_t>(16, (size_t)atoll(v));   // (tests: many regions and pieces on small files)
    if (const char *v = getenv("MC_READER_PIECE_BYTES")) piece_bytes = std::max<size_t>(1, (size_t)atoll(v));
    std::vector<Piece> pieces;
    bool parser_done = false;
    while (!parser_done) {
        double w0 = wt_on ? wnow() : 0;
        st.extend(region_bytes);
        if (wt_on) { const double w1 = wnow(); g_wt[0] += w1 - w0; w0 = w1; }
        if (st.len == 0) {
            if (st.failed) { r_err = "EOFError: compressed file ended before the end-of-stream marker was reached (" + path + ": " + st.prod_msg + ")"; return -3; }
            break;
        }
        // the region: whole lines, unless the stream ends here; a damaged stream never "ends" - what lies in front of the damage is
        // parsed like any region in the middle of a file, and the error surfaces when the parser wants more
        const bool eof = st.at_end && !st.failed;
        size_t e = eof ? st.len : whole_lines(st.win, st.len);
        if (e == 0) {
            if (st.at_end) { r_err = "EOFError: compressed file ended before the end-of-stream marker was reached (" + path + ": " + st.prod_msg + ")"; return -3; }
            region_bytes *= 2; continue;
        }
        const uint8_t *base = st.win;
        // pieces at guessed record starts
        const int np = (int)std::max<size_t>(1, std::min<size_t>(piece_bytes < 4096 ? 4096 : (size_t)T * 4, e / piece_bytes));
        std::vector<size_t> starts{0};
        for (int k = 1; k < np; k++) {
            const size_t g = guess_start(base, e / np * k, e);
            if (g < e && g > starts.back()) starts.push_back(g);
        }
        const size_t npc = starts.size();
        if (pieces.size() < npc) pieces.resize(npc);
        for (size_t k = 0; k < npc; k++) pieces[k].reset();
        for (size_t k = 0; k < npc; k++) { pieces[k].start = starts[k]; pieces[k].stop = k + 1 < npc ? starts[k + 1] : e; }
        if (wt_on) { const double w1 = wnow(); g_wt[1] += w1 - w0; w0 = w1; }
        pool.run((int)npc, [&](int k) { parse_piece(base, e, eof, pieces[k], P); });
        if (wt_on) { const double w1 = wnow(); g_wt[2] += w1 - w0; w0 = w1; }
        // stitch in file order: a piece counts only if the parse so far ended exactly at its (guessed) start; where no piece starts,
        // the parser is continued sequentially up to the next guess
        std::vector<Piece *> order;
        std::vector<std::unique_ptr<Piece>> extra;
        size_t at = 0, k = 0;
        bool done = false;
        while (at < e && !done) {
            while (k < npc && pieces[k].start < at) k++;      // guesses inside a record that ran over them
            Piece *pc;
            if (k < npc && pieces[k].start == at) pc = &pieces[k++];
            else {
                extra.emplace_back(new Piece());
                pc = extra.back().get();
                pc->start = at; pc->stop = k < npc ? pieces[k].start : e;
                parse_piece(base, e, eof, *pc, P);
            }
            done = pc->done;
            if (pc->end == at && pc->recs.empty() && !done) break;      // an unfinished record at the end of the region: the next region starts with it
            order.push_back(pc);
            at = pc->end;
        }
        parser_done = done;
        if (wt_on) { const double w1 = wnow(); g_wt[3] += w1 - w0; w0 = w1; }
        const bool go_on = on_region(order);
        if (wt_on) { const double w1 = wnow(); g_wt[4] += w1 - w0; w0 = w1; }
        if (!go_on) return 0;
        if (parser_done) break;
        if (at == 0) {                                               // one record larger than the region: take more
            if (st.at_end) {
                if (st.failed) { r_err = "EOFError: compressed file ended before the end-of-stream marker was reached (" + path + ": " + st.prod_msg + ")"; return -3; }
                break;
            }
            region_bytes *= 2; continue;
        }
        st.consume(at);
        if (t_peek && region_bytes < full_region) region_bytes = std::min(full_region, region_bytes * 4);
        if (st.len == 0 && st.at_end) {
            if (st.failed) { r_err = "EOFError: compressed file ended before the end-of-stream marker was reached (" + path + ": " + st.prod_msg + ")"; return -3; }
            break;
        }
    }
    return 0;
}

}   // namespace

struct mc_reader {
    std::vector<std::string> paths;
    int32_t L = 0, fastq = 0, qoff = 0, filter_dups = 0;
    int64_t nreads = 0;
    int64_t range_lo = -1, range_hi = -1;                          // mc_reader_open_range
    int64_t bz_b0 = -1, bz_b1 = -1; int bz_kind = 0;               // mc_reader_open_bz2_part
    int64_t gz_k0 = -1, gz_k1 = -1, gz_chunk = 0; int gz_kind = 0; std::unique_ptr<GzPartCtx> gz_ctx;   // mc_reader_open_gz_part
    double min_q = 0, mean_q = 0, max_unknown = 0;
    std::string fasta_out;
    uint8_t *reads = nullptr; size_t reads_cap = 0, reads_n = 0;   // anonymous mapping grown with mremap (no copies, no zero fill up front)
    mc_reader_stats st{};
    WalkTimes wt; double wall = 0;                                 // mc_reader_times
    std::vector<mc_rec_desc> descs; KeptMaps desc_maps;            // mc_reader_describe: the window's records, and the file's mapping they point into (kept until the reader closes)
    // streaming (mc_reader_start / fetch / join): the sampler runs on a thread of its own and publishes how far it has got
    std::thread run_th;
    std::mutex pmu; std::condition_variable pcv;
    int64_t published = 0; bool finished = false; int64_t result = 0; std::string result_err;
    std::shared_mutex buf_mu;                                      // mremap may move the rows while a consumer copies
    ~mc_reader()
    {
        if (run_th.joinable()) run_th.join();
        for (auto &m : desc_maps) munmap((void *)m.first, m.second);
        if (reads) {
            // The buffer of a closed reader is kept for the next one (one buffer, up to the limit mc_reader_trim sets - 4 GB unless told
            // otherwise; a long-lived service is not left holding the 8 GB of one large library): returning gigabytes of pages takes a while
            // (0.17 s for the 3 GB of 20 M reads) and so does faulting them in again - a second run_pipeline() of the same process
            // pays neither.  Anything larger is unmapped, off the caller's time.
            uint8_t *p = reads; const size_t n = reads_cap;
            {
                std::unique_lock<std::mutex> lk(cache_mu());
                if (!cache_ptr() && n <= cache_limit() && !getenv("MC_READER_NO_CACHE")) { cache_ptr() = p; cache_cap() = n; p = nullptr; }
            }
            if (p) { if (n >= ((size_t)64 << 20)) std::thread([p, n] { munmap(p, n); }).detach(); else munmap(p, n); }
        }
    }
    static std::mutex &cache_mu() { static std::mutex m; return m; }
    static uint8_t *&cache_ptr() { static uint8_t *p = nullptr; return p; }
    static size_t &cache_cap() { static size_t n = 0; return n; }
    static size_t &cache_limit() { static size_t n = (size_t)4 << 30; return n; }   // what a closed reader may leave behind (mc_reader_trim; default 4 GB: 20 M reads of 150 bp and some)
    bool reserve(size_t bytes)
    {
        if (bytes <= reads_cap) return true;
        if (!reads) {                                                  // the buffer a closed reader left behind
            std::unique_lock<std::mutex> lk(cache_mu());
            if (cache_ptr()) { std::unique_lock<std::shared_mutex> lk2(buf_mu); reads = cache_ptr(); reads_cap = cache_cap(); cache_ptr() = nullptr; cache_cap() = 0; }
        }
        if (bytes <= reads_cap) return true;
        size_t want = std::max<size_t>(bytes, reads_cap * 2);
        want = (want + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
        std::unique_lock<std::shared_mutex> lk(buf_mu);
        void *p = reads ? mremap(reads, reads_cap, want, MREMAP_MAYMOVE) : mmap(nullptr, want, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (p == MAP_FAILED) return false;
        reads = (uint8_t *)p; reads_cap = want;
        return true;
    }
    void publish(int64_t kept) { std::unique_lock<std::mutex> lk(pmu); published = kept; pcv.notify_all(); }
};

// memcpy of a batch by several threads, a slice each (library-internal: mc_reader_fetch, and mc_search's source in mc_hip.hip).  A batch
// of 2 M reads is 300 - 600 MB: one thread moves that in 40 - 80 ms, and the caller - the uploader of mc_search / mc_search_files, which
// sends the batch to the device next - was what a plain FASTQ waited for end to end (round 6: parser 80 M reads/s, device 69 M reads/s,
// file -> AGS 49 M reads/s).
void mc_host_copy(uint8_t *dst, const uint8_t *src, size_t bytes)
{
    const int nt = bytes >= ((size_t)16 << 20) ? std::max(1, std::min(8, effective_cores() / 2)) : 1;
    if (nt == 1) { memcpy(dst, src, bytes); return; }
    const size_t per = ((bytes / (size_t)nt) + 4095) & ~(size_t)4095;
    std::vector<std::thread> th;
    for (int t = 1; t < nt; t++) { const size_t o = per * (size_t)t; if (o < bytes) th.emplace_back([=] { memcpy(dst + o, src + o, std::min(per, bytes - o)); }); }
    memcpy(dst, src, std::min(per, bytes));
    for (auto &x : th) x.join();
}

extern "C" const char *mc_reader_last_error(void) { return r_err.c_str(); }
extern "C" void mc_set_host_threads(int32_t n) { g_host_threads.store(n > 0 ? n : 0); }

extern "C" mc_reader *mc_reader_open(const char *const *paths, int32_t npaths, int32_t read_len, int64_t nreads, int32_t fastq, int32_t quality_offset,
                                     double min_quality, double mean_quality, double max_unknown, int32_t filter_dups, const char *fasta_out)
{
    if (npaths <= 0 || read_len <= 0 || nreads <= 0) { r_err = "mc_reader_open: bad arguments"; return nullptr; }
    mc_reader *r = new mc_reader();
    for (int i = 0; i < npaths; i++) r->paths.push_back(paths[i]);
    r->L = read_len; r->nreads = nreads; r->fastq = fastq; r->qoff = quality_offset; r->filter_dups = filter_dups;
    r->min_q = min_quality; r->mean_q = mean_quality; r->max_unknown = max_unknown;
    if (fasta_out) r->fasta_out = fasta_out;
    return r;
}

// The sampler on ONE plain (uncompressed, regular) file's byte window [byte_lo, byte_hi): the records that start in it - both ends
// moved to the first record start behind them by the same rule, so consecutive windows cut the file into whole records whoever
// reads them.  What the ranks of a multi-GPU run open, each on its own slice (microbecensus_amd/distributed.py); filter_dups needs
// the whole stream in one place and is not offered.  mc_reader_stats.ragged_end: the window did not end on a record boundary.
extern "C" mc_reader *mc_reader_open_range(const char *path, int64_t byte_lo, int64_t byte_hi, int32_t read_len, int64_t nreads, int32_t fastq, int32_t quality_offset,
                                           double min_quality, double mean_quality, double max_unknown)
{
    if (!path || byte_lo < 0 || byte_hi < byte_lo) { r_err = "mc_reader_open_range: bad arguments"; return nullptr; }
    const char *paths[1] = {path};
    mc_reader *r = mc_reader_open(paths, 1, read_len, nreads, fastq, quality_offset, min_quality, mean_quality, max_unknown, 0, nullptr);
    if (r) { r->range_lo = byte_lo; r->range_hi = byte_hi; }
    return r;
}

// The sampler on the blocks [block_lo, block_hi) of ONE .bz2 file that is well-formed from its first to its last byte (mc_bz2_blocks says how
// many it has): the records that start in the text of those blocks - the blocks of a bzip2 file are independent, so the ranks of a multi-GPU
// run decode and sample their own shares side by side (microbecensus_amd/distributed.py), which a .gz does not allow.  kind: '@' or '>',
// what a record of this file starts with (the first byte of its text).
extern "C" mc_reader *mc_reader_open_bz2_part(const char *path, int64_t block_lo, int64_t block_hi, int32_t kind, int32_t read_len, int64_t nreads, int32_t fastq,
                                              int32_t quality_offset, double min_quality, double mean_quality, double max_unknown)
{
    if (!path || block_lo < 0 || block_hi <= block_lo || (kind != '@' && kind != '>')) { r_err = "mc_reader_open_bz2_part: bad arguments"; return nullptr; }
    const char *paths[1] = {path};
    mc_reader *r = mc_reader_open(paths, 1, read_len, nreads, fastq, quality_offset, min_quality, mean_quality, max_unknown, 0, nullptr);
    if (r) { r->bz_b0 = block_lo; r->bz_b1 = block_hi; r->bz_kind = kind; }
    return r;
}
// blocks of a .bz2 file whose streams all check out (headers, block chain, combined CRCs: mc_pbzip2.h), or -1: cut short, damaged,
// trailing bytes - such a file is read by one sampler (which reports what the reference would)
extern "C" int64_t mc_bz2_blocks(const char *path)
{
    if (!path) { r_err = "null path"; return -1; }
    {
        std::unique_lock<std::mutex> lk(g_bz2_mu);
        if (!g_bz2.load()) { r_err = std::string("cannot load libbz2 for ") + path; return -1; }
    }
    struct stat sb;
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0 || fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size < 14) { if (fd >= 0) ::close(fd); r_err = std::string("cannot map ") + path; return -1; }
    void *m = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (m == MAP_FAILED) { r_err = std::string("cannot map ") + path; return -1; }
    int64_t n = -1;
    try {
        mcbz::Api api;
        api.init = (int (*)(mcbz::BzStreamT *, int, int))g_bz2.init; api.decompress = (int (*)(mcbz::BzStreamT *))g_bz2.decompress; api.end = (int (*)(mcbz::BzStreamT *))g_bz2.end;
        mcbz::ParallelBz2 pz((const uint8_t *)m, (size_t)sb.st_size, api, std::min(std::max(2, reader_threads()), 32));
        if (pz.start(false) && pz.tail_byte == (size_t)sb.st_size) n = (int64_t)pz.blocks.size();
        else r_err = std::string("not a well-formed .bz2 file from its first to its last byte: ") + path;
    } catch (const std::bad_alloc &) { r_err = "out of memory"; }
    munmap(m, (size_t)sb.st_size);
    return n;
}

// ---- a .gz file across the ranks of a multi-GPU run ------------------------------------------------------------------------------------
// A gzip member cannot be ENTERED in the middle - every block may point 32 KB back - but it can be DECODED from the middle speculatively
// (mc_pgzip.h: back-references into the unknown become markers).  So the file is cut into slices of chunks; every rank decodes its
// slice at once; what is sequential is a chain of hand-overs along the slices - where the slice in front ended, and the 32 KB in front of
// that (33 KB a hop) -, after which every rank replaces its markers and samples the records that start in its slice's text.
//   mc_gz_chunks(path, chunk_bytes)            chunks of the file (as the parallel reader cuts it), or -1
//   mc_reader_open_gz_part(path, k0, k1, ...)  the sampler on chunks [k0, k1); run it with mc_reader_start / mc_reader_join
//   mc_reader_gz_provide(r, state, n)          what mc_reader_gz_end_state of the slice in front returned (n = 0: that slice failed); not for k0 = 0
//   mc_reader_gz_end_state(r, out, cap)        waits until the slice is stitched; bytes written (<= 33 KB + 16), or -1
//   mc_reader_gz_finish(r, crc_in, crc_out)    after mc_reader_join: checks the members that end in the slice, given CRC | length (12 bytes) of the
//                                              open member's bytes in front of it (zeros for slice 0); 0, or -3 (gzip.open's "CRC check failed")
extern "C" int64_t mc_gz_chunks(const char *path, int64_t chunk_bytes)
{
    if (!path || chunk_bytes < 4096) { r_err = "bad argument"; return -1; }
    struct stat sb;
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0 || fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size < 18) { if (fd >= 0) ::close(fd); r_err = std::string("cannot map ") + path; return -1; }
    void *m = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (m == MAP_FAILED) { r_err = std::string("cannot map ") + path; return -1; }
    int64_t n = -1;
    { mcgz::ParallelGz pz((const uint8_t *)m, (size_t)sb.st_size, 2, (size_t)chunk_bytes); if (pz.setup()) n = (int64_t)pz.nchunks(); else r_err = std::string("not a gzip file the parallel reader takes: ") + path; }
    munmap(m, (size_t)sb.st_size);
    return n;
}
extern "C" mc_reader *mc_reader_open_gz_part(const char *path, int64_t chunk_lo, int64_t chunk_hi, int64_t chunk_bytes, int32_t kind, int32_t read_len, int64_t nreads,
                                             int32_t fastq, int32_t quality_offset, double min_quality, double mean_quality, double max_unknown)
{
    if (!path || chunk_lo < 0 || chunk_hi <= chunk_lo || chunk_bytes < 4096 || (kind != '@' && kind != '>')) { r_err = "mc_reader_open_gz_part: bad arguments"; return nullptr; }
    const char *paths[1] = {path};
    mc_reader *r = mc_reader_open(paths, 1, read_len, nreads, fastq, quality_offset, min_quality, mean_quality, max_unknown, 0, nullptr);
    if (r) { r->gz_k0 = chunk_lo; r->gz_k1 = chunk_hi; r->gz_chunk = chunk_bytes; r->gz_kind = kind; r->gz_ctx.reset(new GzPartCtx()); }
    return r;
}
static size_t gz_state_pack(const mcgz::ParallelGz::SliceState &st, uint8_t *out, size_t cap)
{
    const size_t need = 16 + st.window.size();
    if (cap < need) return 0;
    memcpy(out, &st.end_bit, 8);
    out[8] = st.member_start; out[9] = st.stop; out[10] = st.bad; out[11] = 0;
    const uint32_t wl = (uint32_t)st.window.size(); memcpy(out + 12, &wl, 4);
    if (wl) memcpy(out + 16, st.window.data(), wl);
    return need;
}
extern "C" int mc_reader_gz_provide(mc_reader *r, const uint8_t *state, int64_t n)
{
    if (!r || !r->gz_ctx) { r_err = "not a .gz part reader"; return -1; }
    GzPartCtx &c = *r->gz_ctx;
    std::unique_lock<std::mutex> lk(c.mu);
    if (!state || n < 16) c.in_fail = true;
    else {
        uint32_t wl = 0; memcpy(&wl, state + 12, 4);
        if ((int64_t)wl + 16 != n || wl > 32768) c.in_fail = true;
        else { memcpy(&c.in.end_bit, state, 8); c.in.member_start = state[8] != 0; c.in.stop = state[9] != 0; c.in.bad = state[10] != 0; c.in.window.assign(state + 16, state + 16 + wl); c.in.ready = true; if (c.in.bad) c.in_fail = true; }
    }
    c.in_ready = true;
    c.cv.notify_all();
    return 0;
}
extern "C" int64_t mc_reader_gz_end_state(mc_reader *r, uint8_t *out, int64_t cap)
{
    if (!r || !r->gz_ctx || !out) { r_err = "not a .gz part reader"; return -1; }
    GzPartCtx &c = *r->gz_ctx;
    std::unique_lock<std::mutex> lk(c.mu);
    c.cv.wait(lk, [&] { return c.out_ready; });
    if (c.out_fail) { r_err = "the slice failed"; return -1; }
    const size_t n = gz_state_pack(c.out, out, (size_t)cap);
    if (!n) { r_err = "buffer too small"; return -1; }
    return (int64_t)n;
}
extern "C" int mc_reader_gz_finish(mc_reader *r, const uint8_t *crc_in, uint8_t *crc_out)
{
    if (!r || !r->gz_ctx || !crc_in || !crc_out) { r_err = "not a .gz part reader"; return -1; }
    uint32_t ci = 0, co = 0; uint64_t li = 0, lo = 0;
    memcpy(&ci, crc_in, 4); memcpy(&li, crc_in + 4, 8);
    if (!mcgz::ParallelGz::finish_crc(r->gz_ctx->segs, r->gz_ctx->nsegs_own, ci, li, &co, &lo)) { r_err = "BadGzipFile: CRC check failed"; return -3; }
    memcpy(crc_out, &co, 4); memcpy(crc_out + 4, &lo, 8);
    return 0;
}

extern "C" void mc_reader_close(mc_reader *r) { delete r; }

static int64_t reader_run(mc_reader *r);
extern "C" int64_t mc_reader_run(mc_reader *r)
{
    if (!r) { r_err = "null reader"; return -1; }
    try { return reader_run(r); }
    catch (const std::bad_alloc &) { r_err = "out of memory in the read sampler"; return -1; }
}
static int64_t reader_run(mc_reader *r)
{
    r->reads_n = 0; memset(&r->st, 0, sizeof r->st);
    r->wt = WalkTimes(); r->wall = 0;
    const double run_t0 = Stream::now();
    struct Wall { mc_reader *r; double t0; ~Wall() {
        r->wall = Stream::now() - t0;
        if (getenv("MC_READER_TIMING")) fprintf(stderr, "reader: %.3f s; its own thread: input %.3f, guesses %.3f, parse %.3f, stitch %.3f, verdicts + places + copies %.3f (of it the walkers of the duplicate classes %.3f)\n",
                                                r->wall, r->wt.v[0], r->wt.v[1], r->wt.v[2], r->wt.v[3], r->wt.v[4], r->wt.v[5]);
    } } wall_guard{r, run_t0};
    FILE *out = nullptr;
    if (!r->fasta_out.empty()) {
        out = fopen(r->fasta_out.c_str(), "w");
        if (!out) { r_err = "cannot write " + r->fasta_out; return -1; }
        setvbuf(out, nullptr, _IOFBF, 1 << 22);
    }
    // The parsers of a compressed input share the CPUs with the inflate workers, which keep theirs busy all the time, and a .gz delivers a
    // sixth of the text a plain file does: half of the usable CPUs are plenty (16 usable CPUs, 12 inflate workers: 32 parsers 10.2 M
    // reads/s, 16: 10.7, 8: 11.2, 4: 11.1), unless the caller capped the threads itself (mc_set_host_threads, MC_READER_THREADS).
    int nparse = reader_threads();
    if (!getenv("MC_READER_THREADS") && g_host_threads.load() < 1) {
        bool packed = false;
        for (const std::string &f : r->paths) packed = packed || has_ext(f.c_str(), ".gz") || has_ext(f.c_str(), ".bz2");
        if (packed) nparse = std::min(nparse, std::max(2, effective_cores() / 2));
    }
    Pool pool(nparse);
    Params P; P.L = (size_t)r->L; P.fastq = r->fastq; P.qoff = r->qoff; P.dups = r->filter_dups;
    const bool serial = getenv("MC_READER_SERIAL_SAMPLER") != nullptr;   // (tests compare the two forms)
    P.decide = !serial && !r->filter_dups;
    P.shards = !serial && r->filter_dups;
    P.max_unknown = r->max_unknown; P.mean_q = r->mean_q; P.min_q = r->min_q;
    SeqSet seen;                                                    // the serial form's set
    std::unique_ptr<SeqSet[]> seen_sh(P.shards ? new SeqSet[NSHARD] : nullptr);
    struct Maps { KeptMaps v; ~Maps() { for (auto &m : v) munmap((void *)m.first, m.second); } } maps;   // the plain files the sets point into             // the sets of the duplicate classes' walkers
    const size_t L = (size_t)r->L;
    int64_t kept = 0, rcode = 0;
    char idbuf[32];
    // what the reference raises at a record (R_ERR), in the order its sampler meets the causes: reverse_complement() of the duplicate
    // test, then the quality filter's rec.phred() of a record without qualities, then mean() of an empty list
    auto err_text = [&](const Rec &rec) -> const char * {
        if (r->filter_dups && (rec.flags & R_RCBAD)) return "KeyError: base outside ACGTN in reverse_complement";
        if (!(rec.flags & R_QUAL)) return "TypeError: record without qualities in a FASTQ run";
        return "ValueError: empty quality string";
    };
    for (const std::string &path : r->paths) {
        t_range_lo = r->range_lo; t_range_hi = r->range_hi;        // (consumed by Stream::open on this thread)
        t_bz_b0 = r->bz_b0; t_bz_b1 = r->bz_b1; t_bz_kind = r->bz_kind;
        t_gz_k0 = r->gz_k0; t_gz_k1 = r->gz_k1; t_gz_chunk = r->gz_chunk; t_gz_kind = r->gz_kind; t_gz_ctx = r->gz_ctx.get();
        struct BzPart { ~BzPart() { t_bz_b0 = t_bz_b1 = -1; t_gz_k0 = t_gz_k1 = -1; t_gz_ctx = nullptr; } } bz_part_guard;
        const int rc = walk_file(path, P, pool, [&](std::vector<Piece *> &order) -> bool {
            const int64_t kept0 = kept;
            bool full = false;
            if (P.shards) {
                // -d (reference :345, :354): a record is a duplicate when its sequence or the reverse complement of it was ACCEPTED before -
                // the test comes before the quality filter, only accepted reads enter the set.  So a record's fate depends on nothing but
                // the earlier records of its own class {s, rc(s)}: in file order inside the class, everything in front of the first record
                // that passes the quality filter fails it, that one is accepted, everything behind it is a duplicate.  The classes are dealt
                // to NSHARD sets by their key - min(hash(s), hash(rc(s))), both computed by the parser's threads - and every set walks ITS
                // records of the region in file order with the very loop of the serial form; the sets live as long as the sampler.  The
                // head-take (:356) only decides how far the verdicts count: below.
                std::atomic<bool> oom{false};
                const double tq0 = Stream::now();
                pool.run(NSHARD, [&](int sh) {
                    SeqSet &set = seen_sh[sh];
                    std::vector<ShItem *> mine;
                    for (Piece *pc : order) for (uint32_t j = pc->sh_off[sh]; j < pc->sh_off[sh + 1]; j++) mine.push_back(&pc->sh_items[j]);
                    const size_t n = mine.size(), AHEAD = 8;
                    for (size_t i = 0; i < n; i++) {
                        if (i + AHEAD < n) { set.prefetch(mine[i + AHEAD]->h1); set.prefetch(mine[i + AHEAD]->h2); }
                        ShItem &it = *mine[i];
                        if (set.contains(it.h1, it.seq, it.len, false) || set.contains(it.h2, it.seq, it.len, true)) { it.q = R_DUP; continue; }
                        if ((it.q & R_PASS) && !set.insert(it.h1, it.seq, it.len, (it.q & R_STABLE) != 0)) { oom.store(true); return; }
                        it.q &= (uint8_t)~R_STABLE;
                    }
                });
                if (oom.load()) { r_err = "out of memory for the set of accepted sequences"; rcode = -1; return false; }
                const double tq1 = Stream::now();
                r->wt.v[5] += tq1 - tq0;
                pool.run((int)order.size(), [&](int k) {
                    Piece *pc = order[k];
                    for (const ShItem &it : pc->sh_items) pc->recs[it.idx].flags |= it.q;
                    for (const Rec &rec : pc->recs) {
                        if (rec.flags & R_SHORT) pc->n_short++;
                        else if (rec.flags & R_ERR) pc->anomaly = true;
                        else if (rec.flags & R_DUP) pc->n_dup++;
                        else if (rec.flags & R_PASS) pc->n_pass++;
                        else pc->n_lowq++;
                    }
                });
            }
            if (P.decide || P.shards) {
                // The verdicts are in: what is left in file order is the place of every piece's accepted reads - a running sum over the
                // pieces - and the piece in which the sample gets full (or which holds a record the reference raises at), walked record by record.
                std::vector<int64_t> base(order.size(), 0), take(order.size(), 0);
                size_t np = 0;
                for (; np < order.size() && !full; np++) {
                    Piece *pc = order[np];
                    if (pc->ragged) r->st.ragged_end = 1;
                    base[np] = kept - kept0;
                    if (pc->anomaly || (r->nreads > 0 && kept + pc->n_pass >= r->nreads)) {   // the sample gets full inside this piece (or with its last accepted read)
                        for (Rec &rec : pc->recs) {
                            r->st.records++; r->st.bases += (int64_t)rec.len;
                            if (rec.flags & R_SHORT) { r->st.too_short++; continue; }
                            if (rec.flags & R_DUP) { r->st.dups++; continue; }
                            if (rec.flags & R_ERR) { r_err = err_text(rec); rcode = -3; return false; }
                            if (!(rec.flags & R_PASS)) { r->st.low_qual++; continue; }
                            take[np]++; kept++;
                            if (kept == r->nreads) { full = true; break; }
                        }
                    } else {
                        r->st.records += (int64_t)pc->recs.size(); r->st.bases += pc->bases; r->st.too_short += pc->n_short; r->st.low_qual += pc->n_lowq; r->st.dups += pc->n_dup;
                        take[np] = pc->n_pass; kept += pc->n_pass;
                    }
                }
                if (kept > kept0) {
                    if (!r->reserve((size_t)kept * L)) { r_err = "out of memory for the sampled reads"; rcode = -1; return false; }
                    uint8_t *dst = r->reads + (size_t)kept0 * L;
                    pool.run((int)np, [&](int k) {
                        int64_t j = base[k]; const int64_t stop = j + take[k];
                        for (const Rec &rec : order[k]->recs) { if (j == stop) break; if (rec.flags & R_PASS) { memcpy(dst + (size_t)j * L, rec.seq, L); j++; } }
                    });
                    if (out) {                                         // the FASTA copy of the sample (reference :352), in order
                        for (int64_t i = kept0; i < kept; i++) {
                            const int k = snprintf(idbuf, sizeof idbuf, ">%lld\n", (long long)i);
                            fwrite(idbuf, 1, (size_t)k, out); fwrite(r->reads + (size_t)i * L, 1, L, out); fputc('\n', out);
                        }
                    }
                    r->publish(kept);
                }
                return !full;
            }
            // the serial form (MC_READER_SERIAL_SAMPLER): the sampler's decisions, record by record in file order
            for (Piece *pc : order) {
                if (pc->ragged) r->st.ragged_end = 1;
                for (Rec &rec : pc->recs) {
                    r->st.records++;
                    r->st.bases += (int64_t)rec.len;
                    if (rec.flags & R_SHORT) { r->st.too_short++; continue; }
                    if (r->filter_dups) {
                        if (seen.contains(rec.h1, rec.seq, rec.len, false)) { r->st.dups++; continue; }
                        if (rec.flags & R_RCBAD) { r_err = "KeyError: base outside ACGTN in reverse_complement"; rcode = -3; return false; }
                        if (seen.contains(rec.h2, rec.seq, rec.len, true)) { r->st.dups++; continue; }
                    }
                    bool fail = (double)(100 * (long long)rec.ncount) / (double)L > r->max_unknown;
                    if (!fail && r->fastq) {
                        if (!(rec.flags & R_QUAL)) { r_err = "TypeError: record without qualities in a FASTQ run"; rcode = -3; return false; }
                        if (rec.nq == 0) { r_err = "ValueError: empty quality string"; rcode = -3; return false; }
                        if ((double)rec.qsum / (double)rec.nq < r->mean_q) fail = true;
                        else if ((double)rec.qmin < r->min_q) fail = true;
                    }
                    if (fail) { r->st.low_qual++; continue; }
                    rec.out = (uint32_t)(kept - kept0);
                    if (out) {
                        const int k = snprintf(idbuf, sizeof idbuf, ">%lld\n", (long long)kept);
                        fwrite(idbuf, 1, (size_t)k, out); fwrite(rec.seq, 1, L, out); fputc('\n', out);
                    }
                    kept++;
                    if (r->filter_dups && !seen.insert(rec.h1, rec.seq, rec.len, false)) { r_err = "out of memory for the set of accepted sequences"; rcode = -1; return false; }
                    if (kept == r->nreads) { full = true; break; }
                }
                if (full) break;
            }
            // the accepted reads of the region -> output rows (copied by the workers, piece by piece)
            if (kept > kept0) {
                if (!r->reserve((size_t)kept * L)) { r_err = "out of memory for the sampled reads"; rcode = -1; return false; }
                uint8_t *dst = r->reads + (size_t)kept0 * L;
                pool.run((int)order.size(), [&](int k) { for (const Rec &rec : order[k]->recs) if (rec.out != ~0u) memcpy(dst + (size_t)rec.out * L, rec.seq, L); });
                r->publish(kept);
            }
            return !full;
        }, P.shards ? &maps.v : nullptr, &r->wt);
        t_range_lo = t_range_hi = -1;
        if (rc < 0 && rcode == 0) rcode = rc;
        if (rcode < 0 || kept == r->nreads) break;
    }
    if (out) fclose(out);
    if (rcode < 0) return rcode;
    r->reads_n = (size_t)kept;
    r->st.exhausted = (kept < r->nreads) ? 1 : 0;             // every file was read to its end: bases is count_bases()
    r->st.sampled = kept;
    return kept;
}

extern "C" const uint8_t *mc_reader_reads(mc_reader *r) { return r ? r->reads : nullptr; }
extern "C" int mc_reader_get_stats(mc_reader *r, mc_reader_stats *out) { if (!r || !out) return -1; *out = r->st; return 0; }

// ---- -d across the ranks of a multi-GPU run (microbecensus_amd/distributed.py, stream_batches_sharded_dups) ---------------------------
// The duplicate rule (process_seqfile :345, :354) is class-local (mc_reader_run), so the expensive part of the sampler - parsing, the
// quality filter, both hashes - can run on every rank's own byte window; what has to be seen in file order is only a 32-byte descriptor
// per record.  mc_reader_describe() makes the descriptors of a window; the ranks exchange them; mc_dupset_walk() gives every record of
// the round its verdict (every rank runs it on the same descriptors: same verdicts everywhere, sequences compared on the file's own
// mapping); mc_reader_take() copies the accepted reads of the rank's own window.
extern "C" int64_t mc_reader_describe(mc_reader *r, const mc_rec_desc **out)
{
    if (!r || !out) { r_err = "bad argument"; return -1; }
    if (r->range_lo < 0 || r->paths.size() != 1) { r_err = "mc_reader_describe: the reader must be opened with mc_reader_open_range"; return -1; }
    try {
        r->descs.clear(); memset(&r->st, 0, sizeof r->st);
        for (auto &m : r->desc_maps) munmap((void *)m.first, m.second);
        r->desc_maps.clear();
        Pool pool(reader_threads());
        Params P; P.L = (size_t)r->L; P.fastq = r->fastq; P.qoff = r->qoff; P.dups = 1;
        P.max_unknown = r->max_unknown; P.mean_q = r->mean_q; P.min_q = r->min_q;
        bool joined = false;
        t_range_lo = r->range_lo; t_range_hi = r->range_hi;
        const int rc = walk_file(r->paths[0], P, pool, [&](std::vector<Piece *> &order) -> bool {
            for (Piece *pc : order) {
                if (pc->ragged) r->st.ragged_end = 1;
                for (const Rec &rec : pc->recs) {
                    r->st.records++; r->st.bases += (int64_t)rec.len;
                    mc_rec_desc d; d.h1 = rec.h1; d.h2 = rec.h2; d.seq_off = (uint64_t)(uintptr_t)rec.seq; d.len = rec.len; d.flags = rec.flags & (R_SHORT | R_QUAL | R_RCBAD);
                    if (!(rec.flags & R_SHORT)) {
                        if (!(rec.flags & R_STABLE)) joined = true;              // (a sequence over several lines lies in the piece's arena, not in the file)
                        Rec t = rec;
                        d.flags |= !decide(t, P) ? R_ERR : (t.flags & (R_PASS | R_LOWQ));
                    }
                    r->descs.push_back(d);
                }
            }
            return true;
        }, &r->desc_maps);
        t_range_lo = t_range_hi = -1;
        if (rc < 0) return rc;
        const uint8_t *base = r->desc_maps.empty() ? nullptr : r->desc_maps[0].first;
        for (mc_rec_desc &d : r->descs) d.seq_off = (d.flags & R_SHORT) || !base ? 0 : (uint64_t)((const uint8_t *)(uintptr_t)d.seq_off - base);
        if (joined) r->st.ragged_end = 1;                                        // (the caller falls back to the sampler on one rank, as for a window that does not end on a record boundary)
        r->st.exhausted = 1;
        *out = r->descs.data();
        return (int64_t)r->descs.size();
    } catch (const std::bad_alloc &) { r_err = "out of memory in the read sampler"; return -1; }
}

struct mc_dupset {
    std::unique_ptr<SeqSet[]> sets{new SeqSet[NSHARD]};
    std::vector<std::pair<std::string, std::pair<const uint8_t *, size_t>>> maps;   // the files the sets point into: mapped until the set closes
    ~mc_dupset() { for (auto &m : maps) if (m.second.first) munmap((void *)m.second.first, m.second.second); }
};
extern "C" mc_dupset *mc_dupset_open(void) { try { return new mc_dupset(); } catch (const std::bad_alloc &) { r_err = "out of memory"; return nullptr; } }
extern "C" void mc_dupset_close(mc_dupset *s) { delete s; }
// verdict[i]: the descriptor's R_SHORT | R_QUAL | R_RCBAD bits and ONE of R_SHORT, R_DUP, R_PASS, R_LOWQ, R_ERR - what the reference's sampler
// would decide at record i given every record walked before (this call's and the earlier calls' on this set), in the order of the array
extern "C" int mc_dupset_walk(mc_dupset *s, const char *path, const mc_rec_desc *d, int64_t n, uint8_t *verdict)
{
    if (!s || !path || (n > 0 && (!d || !verdict)) || n < 0) { r_err = "bad argument"; return -1; }
    try {
        const uint8_t *base = nullptr; size_t size = 0;
        for (auto &m : s->maps) if (m.first == path) { base = m.second.first; size = m.second.second; }
        if (!base) {
            const int fd = ::open(path, O_RDONLY);
            struct stat sb;
            if (fd < 0 || fstat(fd, &sb) != 0) { if (fd >= 0) ::close(fd); r_err = std::string("cannot open ") + path; return -1; }
            size = (size_t)sb.st_size;
            void *m = size ? mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0) : nullptr;
            ::close(fd);
            if (size && m == MAP_FAILED) { r_err = std::string("cannot map ") + path; return -1; }
            base = (const uint8_t *)m;
            s->maps.emplace_back(path, std::make_pair(base, size));
        }
        std::vector<uint32_t> off(NSHARD + 1, 0), idx;
        auto shard_of = [&](const mc_rec_desc &x) { return (unsigned)((x.h1 < x.h2 ? x.h1 : x.h2) >> 58); };
        for (int64_t i = 0; i < n; i++) {
            const uint8_t f = d[i].flags;
            verdict[i] = f & (R_SHORT | R_QUAL | R_RCBAD);
            if (f & R_SHORT) continue;
            if (f & R_RCBAD) { verdict[i] |= R_ERR; continue; }
            if (d[i].seq_off + d[i].len > size) { r_err = "mc_dupset_walk: a descriptor points outside its file"; return -1; }
            off[shard_of(d[i]) + 1]++;
        }
        for (int k = 0; k < NSHARD; k++) off[k + 1] += off[k];
        idx.resize(off[NSHARD]);
        { std::vector<uint32_t> at(off.begin(), off.end() - 1); for (int64_t i = 0; i < n; i++) if (!(d[i].flags & (R_SHORT | R_RCBAD))) idx[at[shard_of(d[i])]++] = (uint32_t)i; }
        std::atomic<bool> oom{false};
        Pool pool(reader_threads());
        pool.run(NSHARD, [&](int sh) {
            SeqSet &set = s->sets[sh];
            const uint32_t a = off[sh], b = off[sh + 1], AHEAD = 8;
            for (uint32_t k = a; k < b; k++) {
                if (k + AHEAD < b) { set.prefetch(d[idx[k + AHEAD]].h1); set.prefetch(d[idx[k + AHEAD]].h2); }
                const mc_rec_desc &x = d[idx[k]];
                const uint8_t *seq = base + x.seq_off;
                if (set.contains(x.h1, seq, x.len, false) || set.contains(x.h2, seq, x.len, true)) { verdict[idx[k]] |= R_DUP; continue; }
                verdict[idx[k]] |= x.flags & (R_PASS | R_LOWQ | R_ERR);
                if ((x.flags & R_PASS) && !set.insert(x.h1, seq, x.len, true)) { oom.store(true); return; }
            }
        });
        if (oom.load()) { r_err = "out of memory for the set of accepted sequences"; return -1; }
        return 0;
    } catch (const std::bad_alloc &) { r_err = "out of memory in the read sampler"; return -1; }
}
// the first read_len bases of the records of the described window whose verdict is R_PASS, in file order, at most max_take of them -> dst
extern "C" int64_t mc_reader_take(mc_reader *r, const uint8_t *verdict, int64_t n, int64_t max_take, uint8_t *dst)
{
    if (!r || n != (int64_t)r->descs.size() || (n > 0 && !verdict) || max_take < 0) { r_err = "mc_reader_take: the verdicts of the described window, one per record"; return -1; }
    const uint8_t *base = r->desc_maps.empty() ? nullptr : r->desc_maps[0].first;
    const size_t L = (size_t)r->L;
    int64_t k = 0;
    for (int64_t i = 0; i < n && k < max_take; i++) if (verdict[i] & R_PASS) { if (dst) memcpy(dst + (size_t)k * L, base + r->descs[(size_t)i].seq_off, L); k++; }
    return k;
}

extern "C" int64_t mc_count_bases(const char *const *paths, int32_t npaths)
{
    int64_t total = 0;
    try {
        Pool pool(reader_threads());
        Params P; P.count_only = true;
        for (int i = 0; i < npaths; i++) {
            const int rc = walk_file(paths[i], P, pool, [&](std::vector<Piece *> &order) -> bool { for (Piece *pc : order) total += pc->bases; return true; });
            if (rc < 0) return rc;
        }
    } catch (const std::bad_alloc &) { r_err = "out of memory in the read sampler"; return -1; }
    return total;
}

// auto_detect_quality_offset (microbe_census.py:175-187): the first quality character of the file that decides - one of
// !"#$%&'()*+,-./0123456789 says 32, one of K..~ says 64, a file without either says 32.  The reference walks the records with
// its Python parser until it finds one (the whole file if every quality lies in ':'..'J'); this walks them with the native
// parser.  Returns 32 or 64; -2 when a record has no quality line (the caller then takes the Python path and fails the way the
// reference does); < 0 otherwise as the reader's other entry points.
extern "C" int32_t mc_quality_offset(const char *path)
{
    if (!path) { r_err = "null path"; return -1; }
    int32_t answer = 0;
    Pool pool(reader_threads());
    Params P; P.count_only = true; P.find_qoff = true;
    struct Peek { Peek() { t_peek = true; } ~Peek() { t_peek = false; } } peek;   // most files decide in their first record
    // (a file whose qualities never decide - simulated reads with a constant 'I' - is walked to its end as the reference walks it: the pieces
    // look for their first deciding character on the parser's threads, in file order only their answers are looked at - 1.05 s for 20 M
    // reads when this loop read every quality itself)
    const int rc = walk_file(path, P, pool, [&](std::vector<Piece *> &order) -> bool {
        for (Piece *pc : order) if (pc->qoff_answer) { answer = pc->qoff_answer; return false; }
        return true;
    });
    if (rc < 0) return rc;
    return answer ? answer : 32;
}

// ---- streaming form ---------------------------------------------------------------------------------------------------
extern "C" int mc_reader_start(mc_reader *r)
{
    if (!r) { r_err = "null reader"; return -1; }
    if (r->run_th.joinable()) { r_err = "the sampler is already running"; return -1; }
    { std::unique_lock<std::mutex> lk(r->pmu); r->published = 0; r->finished = false; r->result = 0; }
    r->run_th = std::thread([r] {
        const int64_t rc = mc_reader_run(r);
        std::unique_lock<std::mutex> lk(r->pmu);
        r->result = rc; r->result_err = r_err; r->finished = true;
        if (rc > r->published) r->published = rc;
        r->pcv.notify_all();
    });
    return 0;
}

extern "C" int64_t mc_reader_fetch(mc_reader *r, int64_t first, int64_t max_reads, uint8_t *dst)
{
    if (!r || first < 0 || max_reads < 0) { r_err = "bad argument"; return -1; }
    int64_t have;
    {
        std::unique_lock<std::mutex> lk(r->pmu);
        r->pcv.wait(lk, [&] { return r->finished || r->published >= first + max_reads; });
        if (r->finished && r->result < 0) { r_err = r->result_err; return r->result; }
        have = r->published;
    }
    const int64_t n = std::max<int64_t>(0, std::min(max_reads, have - first));
    if (n > 0 && dst) {
        std::shared_lock<std::shared_mutex> lk(r->buf_mu);
        mc_host_copy(dst, r->reads + (size_t)first * (size_t)r->L, (size_t)n * (size_t)r->L);
    }
    return n;
}

extern "C" int64_t mc_reader_join(mc_reader *r)
{
    if (!r) { r_err = "null reader"; return -1; }
    if (r->run_th.joinable()) r->run_th.join();
    if (r->result < 0) r_err = r->result_err;
    return r->result;
}

// Releases the buffer a closed reader left for the next one, and sets how large a buffer may be kept from now on (keep_bytes;
// 0: none is kept).  For long-lived processes that sample a large library once.
extern "C" void mc_reader_trim(int64_t keep_bytes)
{
    uint8_t *p = nullptr; size_t n = 0;
    {
        std::unique_lock<std::mutex> lk(mc_reader::cache_mu());
        mc_reader::cache_limit() = keep_bytes > 0 ? (size_t)keep_bytes : 0;
        if (mc_reader::cache_ptr() && mc_reader::cache_cap() > mc_reader::cache_limit()) { p = mc_reader::cache_ptr(); n = mc_reader::cache_cap(); mc_reader::cache_ptr() = nullptr; mc_reader::cache_cap() = 0; }
    }
    if (p) munmap(p, n);
}
// Seconds of the last mc_reader_run by phase: [0] the whole run; on the sampler's own thread: [1] waiting for input (inflate), [2] guessing the
// pieces' starts, [3] the parse on all workers, [4] stitching, [5] verdicts + places + copies, [6] of it the walkers of the duplicate classes.
extern "C" int32_t mc_reader_times(const mc_reader *r, double *out, int32_t n)
{
    if (!r || !out) return -1;
    const double v[7] = {r->wall, r->wt.v[0], r->wt.v[1], r->wt.v[2], r->wt.v[3], r->wt.v[4], r->wt.v[5]};
    int32_t k = 0;
    for (; k < n && k < 7; k++) out[k] = v[k];
    return k;
}
extern "C" int32_t mc_reader_read_len(const mc_reader *r) { return r ? r->L : 0; }
extern "C" int64_t mc_reader_nreads(const mc_reader *r) { return r ? r->nreads : 0; }
