// mc_reader.cpp - native read sampler: the host stage in front of the search (include/mcensus.h, mc_reader_*).
//
// Replaces, with identical results, the Python stages of the reference that feed the hot path
//   open_file            /root/reference/microbe_census/microbe_census.py:47-59   (plain and .gz; .bz2 stays in Python)
//   parse_seqs           :294-325   readfq-style FASTA/FASTQ records, with its quirks (below)
//   quality_filter       :265-279
//   process_seqfile      :328-367   head-take sampler: files in order, records in order
//   count_bases          :573-584   second pass over every file
// and hands the accepted reads over as one packed n x read_len byte matrix, which is what mc_search() takes.
//
// Quirks of the reference reproduced here:
//   * text mode with universal newlines: "\r\n" and a lone "\r" end a line like "\n";
//   * every line loses exactly its last character - the newline - so the last line of a file that does not end in a
//     newline loses its last real character;
//   * the record name is the header without its first character up to the first SPACE (tabs stay);
//   * sequence lines run until a line that starts with '@', '+' or '>'; after '+', quality lines are consumed until they
//     cover the sequence length; a file that ends inside the qualities yields the record without qualities and stops;
//   * a read is too short when len(seq) < L; duplicates (the untrimmed sequence or its reverse complement already
//     accepted) are tested BEFORE the quality filter and only when requested; only accepted reads enter the set;
//     reverse_complement knows ACGTN only - any other character is an error (KeyError in the reference);
//   * QC looks at the first L bases / qualities: 100*count('N')/L > max_unknown, mean(q) < mean_quality, min(q) < min_quality
//     with q = ord(c) - quality_offset, in IEEE double like numpy.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <zlib.h>

#include "../../include/mcensus.h"

namespace {

thread_local std::string r_err;

struct LineSource {   // bytes of a plain or gzip file, split into lines with universal-newline semantics
    // A producer thread inflates / reads the file 4 MB at a time into a ring of three blocks while the caller parses: for
    // .gz input the inflate (the slower half) runs beside the parser instead of in front of it.
    enum { NBLK = 3, BLK = 1 << 22 };
    gzFile gz = nullptr;
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::vector<unsigned char> ring[NBLK];
    size_t ring_n[NBLK] = {0, 0, 0};
    int head = 0, tail = 0, count = 0;
    bool prod_done = false, stop = false, prod_err = false;   // prod_err: the stream is truncated or corrupt (gzip.open raises there)
    std::string prod_msg;
    std::vector<unsigned char> buf;
    size_t pos = 0, end = 0;
    bool eof = false, pending_cr = false;
    ~LineSource() { close(); }
    bool open(const char *path)
    {
        gz = gzopen(path, "rb");                 // zlib reads plain files transparently as well
        if (!gz) return false;
        gzbuffer(gz, 1 << 20);
        buf.resize(BLK);
        for (auto &r : ring) r.resize(BLK);
        th = std::thread([this] { produce(); });
        return true;
    }
    void produce()
    {
        for (;;) {
            int slot;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [this] { return count < NBLK || stop; });
                if (stop) return;
                slot = tail;
            }
            const int n = gzread(gz, ring[slot].data(), (unsigned)BLK);
            std::unique_lock<std::mutex> lk(mu);
            if (n <= 0) {
                // gzip.open raises EOFError / BadGzipFile on a truncated or corrupt stream (reference :47-59); zlib reports a
                // truncated stream as Z_BUF_ERROR, a damaged one as Z_DATA_ERROR - a clean end leaves Z_OK / Z_STREAM_END and gzeof
                int errnum = 0;
                const char *msg = gzerror(gz, &errnum);
                if (n < 0 || (errnum != Z_OK && errnum != Z_STREAM_END) || !gzeof(gz)) { prod_err = true; prod_msg = msg ? msg : "read error"; }
                prod_done = true; cv.notify_all(); return;
            }
            ring_n[slot] = (size_t)n; tail = (tail + 1) % NBLK; count++;
            cv.notify_all();
        }
    }
    void close()
    {
        if (th.joinable()) {
            { std::unique_lock<std::mutex> lk(mu); stop = true; cv.notify_all(); }
            th.join();
        }
        if (gz) { gzclose(gz); gz = nullptr; }
    }
    bool fill()
    {
        if (eof) return false;
        if (pos < end) { memmove(buf.data(), buf.data() + pos, end - pos); }
        end -= pos; pos = 0;
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [this] { return count > 0 || prod_done; });
        if (count == 0) { eof = true; return false; }
        const size_t n = ring_n[head];
        if (end + n > buf.size()) buf.resize(end + n > buf.size() * 2 ? end + n : buf.size() * 2);
        lk.unlock();
        memcpy(buf.data() + end, ring[head].data(), n);          // (the producer never touches a block that is still counted)
        end += n;
        lk.lock();
        head = (head + 1) % NBLK; count--;
        cv.notify_all();
        return true;
    }
    // Next line: [*p, *p + *n) = the characters in front of the terminator (valid until the next call); *nl = the line
    // had a terminator.  Returns false at end of file (no line).
    bool next(const unsigned char **p, size_t *n, bool *nl)
    {
        if (pending_cr) {                        // "\r" ended the previous line: a directly following "\n" belongs to it
            if (pos == end) fill();
            if (pos < end && buf[pos] == '\n') pos++;
            pending_cr = false;
        }
        size_t scan = 0;                         // characters after pos already known to hold no terminator
        for (;;) {
            const unsigned char *b = buf.data() + pos;
            const size_t avail = end - pos;
            const unsigned char *q = (const unsigned char *)memchr(b + scan, '\n', avail - scan);
            const size_t lim = q ? (size_t)(q - b) : avail;
            const unsigned char *c = (const unsigned char *)memchr(b + scan, '\r', lim - scan);
            if (c) q = c;
            if (q) {
                const size_t at = (size_t)(q - b);
                *p = b; *n = at; *nl = true;
                pending_cr = (*q == '\r');
                pos += at + 1;
                return true;
            }
            scan = avail;
            if (!fill()) {
                if (end == pos) return false;
                *p = buf.data() + pos; *n = end - pos; *nl = false;
                pos = end;
                return true;
            }
        }
    }
};

struct Record { std::string seq, qual; bool has_qual = false; };

// parse_seqs (reference :294-325) as a pull parser
struct Parser {
    LineSource src;
    bool have_pending = false, done = false;
    std::string pending;                         // header line without its last character
    static void chomp(const unsigned char *p, size_t n, bool nl, std::string &out)
    { // line[:-1]
        if (!nl && n > 0) n--;
        out.assign((const char *)p, n);
    }
    bool next(Record &rec)
    {
        if (done) return false;
        const unsigned char *p; size_t n; bool nl;
        if (!have_pending) {
            for (;;) {
                if (!src.next(&p, &n, &nl)) { done = true; return false; }
                if (n > 0 && (p[0] == '>' || p[0] == '@')) { chomp(p, n, nl, pending); have_pending = true; break; }
            }
            if (pending.empty()) { done = true; return false; }        // a lone '>' / '@' without newline at the end of the file: `if not last: break`
        }
        have_pending = false;
        rec.seq.clear(); rec.qual.clear(); rec.has_qual = false;
        bool got_next = false;
        for (;;) {
            if (!src.next(&p, &n, &nl)) break;
            if (n > 0 && (p[0] == '@' || p[0] == '+' || p[0] == '>')) { chomp(p, n, nl, pending); got_next = true; break; }
            size_t m = (!nl && n > 0) ? n - 1 : n;
            rec.seq.append((const char *)p, m);
        }
        if (!got_next) { done = true; return true; }                 // last record of the file, no qualities
        if (pending.empty()) { done = true; return true; }           // `not last or last[0] != '+'` short-circuits on the empty header: the record goes out without qualities, the parser stops
        if (pending[0] != '+') { have_pending = true; return true; }
        size_t got = 0;
        bool complete = false;
        for (;;) {
            if (!src.next(&p, &n, &nl)) break;
            size_t m = (!nl && n > 0) ? n - 1 : n;
            rec.qual.append((const char *)p, m);
            // the reference counts len(line) - 1: a last line without newline that is empty cannot occur
            got += m;
            if (got >= rec.seq.size()) { complete = true; break; }
        }
        if (complete) { rec.has_qual = true; return true; }
        rec.qual.clear(); done = true;                                 // truncated qualities: record without them, then stop
        return true;
    }
};

// Exact set of the accepted (untrimmed) sequences for -d: the strings live back to back in one arena, an open-addressing
// table holds (hash, offset) - no allocation per sequence, one cache miss per lookup; equality is decided on the bytes.
struct SeqSet {
    std::vector<char> arena;
    std::vector<uint64_t> hash, off;             // off: arena offset + 1 (0 = empty slot); the length sits in front of the bytes
    size_t used = 0;
    static uint64_t h64(const char *p, size_t n)
    {
        uint64_t h = 0x9E3779B97F4A7C15ull ^ (n * 0xff51afd7ed558ccdull);
        size_t i = 0;
        for (; i + 8 <= n; i += 8) { uint64_t w; memcpy(&w, p + i, 8); h = (h ^ w) * 0x9FB21C651E98DF25ull; h ^= h >> 29; }
        uint64_t w = 0;
        if (i < n) { memcpy(&w, p + i, n - i); h = (h ^ w) * 0x9FB21C651E98DF25ull; h ^= h >> 29; }
        h *= 0xc4ceb9fe1a85ec53ull; h ^= h >> 32;
        return h;
    }
    void grow()
    {
        const size_t cap = hash.empty() ? (1u << 16) : hash.size() * 2;
        std::vector<uint64_t> nh(cap, 0), no(cap, 0);
        for (size_t i = 0; i < hash.size(); i++) if (off[i]) { size_t j = hash[i] & (cap - 1); while (no[j]) j = (j + 1) & (cap - 1); nh[j] = hash[i]; no[j] = off[i]; }
        hash.swap(nh); off.swap(no);
    }
    bool contains(const std::string &s) const
    {
        if (hash.empty()) return false;
        const uint64_t h = h64(s.data(), s.size());
        for (size_t j = h & (hash.size() - 1);; j = (j + 1) & (hash.size() - 1)) {
            if (!off[j]) return false;
            if (hash[j] == h) {
                const char *q = arena.data() + (off[j] - 1);
                uint64_t len; memcpy(&len, q, 8);
                if (len == s.size() && memcmp(q + 8, s.data(), s.size()) == 0) return true;
            }
        }
    }
    void insert(const std::string &s)              // (the caller has checked that it is not there)
    {
        if ((used + 1) * 2 > hash.size()) grow();
        const uint64_t h = h64(s.data(), s.size()), len = s.size();
        const size_t o = arena.size();
        arena.resize(o + 8 + s.size());
        memcpy(arena.data() + o, &len, 8); memcpy(arena.data() + o + 8, s.data(), s.size());
        size_t j = h & (hash.size() - 1);
        while (off[j]) j = (j + 1) & (hash.size() - 1);
        hash[j] = h; off[j] = o + 1; used++;
    }
};

bool is_bz2(const char *path) { size_t n = strlen(path); return n >= 4 && strcmp(path + n - 4, ".bz2") == 0; }

}   // namespace

struct mc_reader {
    std::vector<std::string> paths;
    int32_t L = 0, fastq = 0, qoff = 0, filter_dups = 0;
    int64_t nreads = 0;
    double min_q = 0, mean_q = 0, max_unknown = 0;
    std::string fasta_out;
    std::vector<uint8_t> reads;
    mc_reader_stats st{};
};

extern "C" const char *mc_reader_last_error(void) { return r_err.c_str(); }

extern "C" mc_reader *mc_reader_open(const char *const *paths, int32_t npaths, int32_t read_len, int64_t nreads, int32_t fastq, int32_t quality_offset,
                                     double min_quality, double mean_quality, double max_unknown, int32_t filter_dups, const char *fasta_out)
{
    if (npaths <= 0 || read_len <= 0 || nreads <= 0) { r_err = "mc_reader_open: bad arguments"; return nullptr; }
    for (int i = 0; i < npaths; i++) if (is_bz2(paths[i])) { r_err = "bz2 input is read by the Python stage"; return nullptr; }
    mc_reader *r = new mc_reader();
    for (int i = 0; i < npaths; i++) r->paths.push_back(paths[i]);
    r->L = read_len; r->nreads = nreads; r->fastq = fastq; r->qoff = quality_offset; r->filter_dups = filter_dups;
    r->min_q = min_quality; r->mean_q = mean_quality; r->max_unknown = max_unknown;
    if (fasta_out) r->fasta_out = fasta_out;
    return r;
}

extern "C" void mc_reader_close(mc_reader *r) { delete r; }

static bool revcomp(const std::string &s, std::string &out)
{ // Sequence.reverse_complement (reference :288-292): ACGTN only, anything else is a KeyError there
    static const struct Tab { unsigned char t[256]; Tab() { memset(t, 0, sizeof t); t['A'] = 'T'; t['T'] = 'A'; t['G'] = 'C'; t['C'] = 'G'; t['N'] = 'N'; } } tab;
    const size_t n = s.size();
    out.resize(n);
    const unsigned char *p = (const unsigned char *)s.data() + n;
    char *o = &out[0];
    unsigned char all = 0xFF;
    for (size_t i = 0; i < n; i++) { const unsigned char d = tab.t[*--p]; o[i] = (char)d; all &= (unsigned char)(d ? 0xFF : 0); }
    return all != 0 || n == 0;
}

extern "C" int64_t mc_reader_run(mc_reader *r)
{
    if (!r) { r_err = "null reader"; return -1; }
    r->reads.clear(); memset(&r->st, 0, sizeof r->st);
    FILE *out = nullptr;
    if (!r->fasta_out.empty()) {
        out = fopen(r->fasta_out.c_str(), "w");
        if (!out) { r_err = "cannot write " + r->fasta_out; return -1; }
        setvbuf(out, nullptr, _IOFBF, 1 << 22);
    }
    SeqSet seen;
    std::string rc;
    const size_t L = (size_t)r->L;
    int64_t kept = 0;
    Record rec;
    char idbuf[32];
    int64_t rcode = 0;
    for (const std::string &path : r->paths) {
        Parser ps;
        if (!ps.src.open(path.c_str())) { r_err = "cannot open " + path; rcode = -1; break; }
        while (ps.next(rec)) {
            r->st.records++;
            r->st.bases += (int64_t)rec.seq.size();
            if (rec.seq.size() < L) { r->st.too_short++; continue; }
            if (r->filter_dups) {
                if (seen.contains(rec.seq)) { r->st.dups++; continue; }
                if (!revcomp(rec.seq, rc)) { r_err = "KeyError: base outside ACGTN in reverse_complement"; rcode = -3; break; }
                if (seen.contains(rc)) { r->st.dups++; continue; }
            }
            // quality_filter
            size_t ncount = 0;
            for (size_t i = 0; i < L; i++) ncount += (rec.seq[i] == 'N');
            bool fail = (double)(100 * (long long)ncount) / (double)L > r->max_unknown;
            if (!fail && r->fastq) {
                if (!rec.has_qual) { r_err = "TypeError: record without qualities in a FASTQ run"; rcode = -3; break; }
                size_t nq = rec.qual.size() < L ? rec.qual.size() : L;
                long long sum = 0; int mn = 1 << 30;
                for (size_t i = 0; i < nq; i++) { int q = (int)(unsigned char)rec.qual[i] - r->qoff; sum += q; if (q < mn) mn = q; }
                if (nq == 0) { r_err = "ValueError: empty quality string"; rcode = -3; break; }
                if ((double)sum / (double)nq < r->mean_q) fail = true;
                else if ((double)mn < r->min_q) fail = true;
            }
            if (fail) { r->st.low_qual++; continue; }
            if (out) {
                int k = snprintf(idbuf, sizeof idbuf, ">%lld\n", (long long)kept);
                fwrite(idbuf, 1, (size_t)k, out); fwrite(rec.seq.data(), 1, L, out); fputc('\n', out);
            }
            r->reads.insert(r->reads.end(), rec.seq.begin(), rec.seq.begin() + (long)L);
            kept++;
            if (r->filter_dups) seen.insert(rec.seq);
            if (kept == r->nreads) break;
        }
        if (ps.src.prod_err && rcode == 0 && kept < r->nreads) { r_err = "EOFError: compressed file ended before the end-of-stream marker was reached (" + path + ": " + ps.src.prod_msg + ")"; rcode = -3; }
        ps.src.close();
        if (rcode < 0 || kept == r->nreads) break;
    }
    if (out) fclose(out);
    if (rcode < 0) return rcode;
    r->st.exhausted = (kept < r->nreads) ? 1 : 0;             // every file was read to its end: bases is count_bases()
    r->st.sampled = kept;
    return kept;
}

extern "C" const uint8_t *mc_reader_reads(mc_reader *r) { return r ? r->reads.data() : nullptr; }
extern "C" int mc_reader_get_stats(mc_reader *r, mc_reader_stats *out) { if (!r || !out) return -1; *out = r->st; return 0; }

extern "C" int64_t mc_count_bases(const char *const *paths, int32_t npaths)
{
    int64_t total = 0;
    Record rec;
    for (int i = 0; i < npaths; i++) {
        if (is_bz2(paths[i])) { r_err = "bz2 input is read by the Python stage"; return -1; }
        Parser ps;
        if (!ps.src.open(paths[i])) { r_err = std::string("cannot open ") + paths[i]; return -1; }
        while (ps.next(rec)) total += (int64_t)rec.seq.size();
        const bool bad = ps.src.prod_err;
        const std::string msg = ps.src.prod_msg;
        ps.src.close();
        if (bad) { r_err = std::string("EOFError: compressed file ended before the end-of-stream marker was reached (") + paths[i] + ": " + msg + ")"; return -3; }
    }
    return total;
}
