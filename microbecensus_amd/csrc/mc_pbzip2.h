// mc_pbzip2.h - a .bz2 input decoded by several threads (open_file, /root/reference/microbe_census/microbe_census.py:55-58: Python's bz2
// module; csrc/mc_reader.cpp binds libbz2's low-level interface at run time).  Host code, no GPU.
//
// bzip2 decodes at 15 - 20 MB/s of text per core - one stream of it fed the sampler 0.03 M records/s of 300 bp where a plain file gives 27 -,
// but unlike deflate its BLOCKS are independent: up to 900 KB of text each, Burrows-Wheeler transformed and Huffman coded by itself, framed by a
// 48-bit magic (0x314159265359) and its own CRC-32; a stream is "BZh" + level, blocks, an end magic (0x177245385090) and the stream's CRC (the
// blocks' CRCs combined).  The blocks are bit-aligned, which is why the format's own decoder cannot enter in the middle - but a block cut out
// and shifted to a byte boundary behind a stream header of its own, with an end magic and its own CRC as the stream's behind it, IS a complete
// one-block stream that libbz2 decodes (how lbzip2 and pbzip2 decode foreign files in parallel).  So:
//   * the mapped file is searched for the two magics at every bit offset (by the workers, a slice each);
//   * the candidates are walked in order into streams: header at a byte offset, first magic 32 bits behind it, every block's end = the next
//     candidate, the end magic's CRC == the combination of the block CRCs read off the block headers.  Streams that check out are WELL-FORMED:
//     their blocks are decoded by the workers (a window of blocks ahead of the consumer) and delivered in order;
//   * the first stream that does not check out - cut short, damaged, trailing bytes that are no stream - and everything behind it is left
//     to the ONE-stream decoder of mc_reader.cpp from that stream's first byte (its rules: a clean end between streams, trailing bytes that do
//     not start a stream ignored as Python ignores them, EOFError / "invalid data stream" otherwise - unchanged);
//   * a block of a well-formed stream that libbz2 refuses (a magic that was payload, damage the CRCs in the headers do not show) sends the
//     consumer back to that stream's first byte with the one-stream decoder, skipping what was already delivered of it.
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace mcbz {

struct BzStreamT {      // bz_stream of bzlib.h 1.0
    char *next_in; unsigned int avail_in, total_in_lo32, total_in_hi32;
    char *next_out; unsigned int avail_out, total_out_lo32, total_out_hi32;
    void *state;
    void *(*bzalloc)(void *, int, int); void (*bzfree)(void *, void *); void *opaque;
};
struct Api { int (*init)(BzStreamT *, int, int); int (*decompress)(BzStreamT *); int (*end)(BzStreamT *); };

static const uint64_t MAGIC_BLOCK = 0x314159265359ull, MAGIC_END = 0x177245385090ull;

struct Cand { uint64_t bit; bool end; };
struct Block { uint64_t bit0, bit1; uint32_t crc; int level; size_t stream; };
struct StreamInfo { size_t byte0 = 0, byte_end = 0; int level = 0; size_t first_block = 0, nblocks = 0; };

inline uint64_t bits_at(const uint8_t *p, size_t n, uint64_t bit, int count)   // count <= 56 bits, MSB first; bits behind the data read as 0
{
    uint64_t v = 0;
    const size_t b = (size_t)(bit >> 3);
    for (int k = 0; k < 8; k++) v = (v << 8) | (b + (size_t)k < n ? p[b + (size_t)k] : 0);
    const int s = (int)(bit & 7);
    return (v << s) >> (64 - count);
}

// all occurrences of the two magics in bytes [lo, hi) of the file (a match may end up to 6 bytes behind hi)
inline void scan_magics(const uint8_t *p, size_t n, size_t lo, size_t hi, std::vector<Cand> &out)
{
    uint64_t pat[2][8], mask[8];
    for (int s = 0; s < 8; s++) { mask[s] = 0xFFFFFFFFFFFFull << s; pat[0][s] = MAGIC_BLOCK << s; pat[1][s] = MAGIC_END << s; }
    uint64_t w = 0;
    const size_t from = lo >= 7 ? lo - 7 : 0, to = hi + 6 < n ? hi + 6 : n;
    for (size_t i = from; i < to; i++) {
        w = (w << 8) | p[i];
        if (i < from + 6) continue;                                 // (fewer than 7 bytes in the window: nothing to compare yet)
        for (int s = 0; s < 8; s++) {
            const uint64_t x = w & mask[s];
            if (x == pat[0][s] || x == pat[1][s]) {
                if (i + 1 < 7 && s + 48 > (int)(8 * (i + 1))) continue;
                const uint64_t bit = (uint64_t)(i + 1) * 8 - (uint64_t)s - 48;
                if ((bit >> 3) >= lo && (bit >> 3) < hi) out.push_back(Cand{bit, x == pat[1][s]});
            }
        }
    }
}

struct ParallelBz2 {
    const uint8_t *base; size_t size;
    Api api; int nthreads;
    std::vector<Block> blocks; std::vector<StreamInfo> streams;
    size_t tail_byte = 0;                                           // first byte the one-stream decoder has to take over at (size: nothing left)
    // decoding
    struct Slot { std::vector<uint8_t> out; int state = 0; /* 0 idle, 1 done, 2 failed */ };
    std::vector<std::unique_ptr<Slot>> slots;                       // one per block
    std::vector<std::thread> workers;
    std::mutex mu; std::condition_variable cv_work, cv_done;
    size_t next_decode = 0, limit_decode = 0, next_consume = 0; size_t cur_off = 0;
    size_t blk_end = 0;                                             // blocks [first, blk_end) are decoded (start(): all of them; start_part(): a rank's share)
    bool quit = false;
    // what the consumer is told when the parallel part is over
    bool handover = false; size_t handover_byte = 0; uint64_t handover_skip = 0; bool handover_got_any = false;
    uint64_t delivered_in_stream = 0;

    ParallelBz2(const uint8_t *b, size_t n, const Api &a, int threads) : base(b), size(n), api(a), nthreads(threads < 1 ? 1 : threads) {}
    ~ParallelBz2()
    {
        { std::unique_lock<std::mutex> lk(mu); quit = true; cv_work.notify_all(); }
        for (auto &t : workers) t.join();
    }

    // after start(false): the workers decode blocks [b0, b1) only (a rank of a multi-GPU run decodes its share of the file: the blocks are
    // independent); read_block() then hands them over one by one
    void run_part(size_t b0, size_t b1)
    {
        next_decode = next_consume = b0; blk_end = b1; limit_decode = b0 + (size_t)nthreads * 3;
        for (int t = 0; t < nthreads; t++) workers.emplace_back([this] { work(); });
    }
    bool read_block(std::vector<uint8_t> &dst)                       // appends the next block's text; false: it does not decode, or none is left
    {
        if (next_consume >= blk_end) return false;
        Slot &s = *slots[next_consume];
        {
            std::unique_lock<std::mutex> lk(mu);
            if (limit_decode < next_consume + (size_t)nthreads * 3) { limit_decode = next_consume + (size_t)nthreads * 3; cv_work.notify_all(); }
            cv_done.wait(lk, [&] { return s.state != 0; });
        }
        if (s.state == 2) return false;
        dst.insert(dst.end(), s.out.begin(), s.out.end());
        std::vector<uint8_t>().swap(s.out);
        next_consume++;
        return true;
    }
    // scans and walks the file; false: not even the first stream is well-formed (the caller uses the one-stream decoder for everything)
    bool start(bool run = true)
    {
        if (size < 14 || memcmp(base, "BZh", 3) != 0) return false;
        std::vector<std::vector<Cand>> part((size_t)nthreads);
        {
            std::vector<std::thread> th;
            const size_t per = (size + (size_t)nthreads - 1) / (size_t)nthreads;
            for (int t = 0; t < nthreads; t++) th.emplace_back([&, t] { const size_t lo = per * (size_t)t, hi = lo + per < size ? lo + per : size; if (lo < hi) scan_magics(base, size, lo, hi, part[(size_t)t]); });
            for (auto &x : th) x.join();
        }
        std::vector<Cand> c;
        for (auto &v : part) c.insert(c.end(), v.begin(), v.end());
        std::sort(c.begin(), c.end(), [](const Cand &a, const Cand &b) { return a.bit < b.bit; });
        // the walk: stream by stream while they check out
        size_t ci = 0, pos = 0;
        while (pos + 14 <= size) {
            if (memcmp(base + pos, "BZh", 3) != 0 || base[pos + 3] < '1' || base[pos + 3] > '9') break;
            StreamInfo si; si.byte0 = pos; si.level = base[pos + 3] - '0'; si.first_block = blocks.size();
            while (ci < c.size() && c[ci].bit < (uint64_t)pos * 8 + 32) ci++;
            if (ci >= c.size() || c[ci].bit != (uint64_t)pos * 8 + 32) break;
            uint32_t comb = 0;
            bool ok = false;
            size_t k = ci;
            for (; k < c.size(); k++) {
                if (c[k].end) {
                    const uint32_t scrc = (uint32_t)bits_at(base, size, c[k].bit + 48, 32);
                    if (c[k].bit + 80 > (uint64_t)size * 8 || scrc != comb) break;
                    si.byte_end = (size_t)((c[k].bit + 80 + 7) >> 3);
                    ok = true;
                    break;
                }
                if (k + 1 >= c.size()) break;                       // a block without an end behind it: the stream is cut short
                Block b; b.bit0 = c[k].bit; b.bit1 = c[k + 1].bit; b.crc = (uint32_t)bits_at(base, size, c[k].bit + 48, 32); b.level = si.level; b.stream = streams.size();
                if (b.bit1 - b.bit0 < 80) break;
                comb = ((comb << 1) | (comb >> 31)) ^ b.crc;
                blocks.push_back(b);
            }
            if (!ok) { blocks.resize(si.first_block); break; }
            si.nblocks = blocks.size() - si.first_block;
            streams.push_back(si);
            ci = k + 1; pos = si.byte_end;
        }
        tail_byte = streams.empty() ? 0 : streams.back().byte_end;
        if (streams.empty()) return false;
        slots.resize(blocks.size());
        for (auto &s : slots) s.reset(new Slot());
        blk_end = blocks.size();
        if (!run) return true;
        limit_decode = (size_t)nthreads * 3;
        for (int t = 0; t < nthreads; t++) workers.emplace_back([this] { work(); });
        return true;
    }

    // one block as a stream of its own -> text; false: libbz2 refuses it
    bool decode_block(const Block &b, std::vector<uint8_t> &out) const
    {
        const uint64_t nbits = b.bit1 - b.bit0;
        std::vector<uint8_t> in((size_t)(4 + (nbits + 80 + 7) / 8 + 8), 0);
        in[0] = 'B'; in[1] = 'Z'; in[2] = 'h'; in[3] = (uint8_t)('0' + b.level);
        {   // the block's bits, shifted to the byte boundary behind the header
            const size_t sb = (size_t)(b.bit0 >> 3); const int s = (int)(b.bit0 & 7);
            const size_t nbytes = (size_t)((nbits + 7) >> 3);
            for (size_t i = 0; i < nbytes; i++) {
                const uint32_t hi = sb + i < size ? base[sb + i] : 0, lo = sb + i + 1 < size ? base[sb + i + 1] : 0;
                in[4 + i] = (uint8_t)(((hi << 8 | lo) << s) >> 8);
            }
            const int tail = (int)(nbits & 7);                       // bits of the last byte that belong to the block
            if (tail) in[4 + nbytes - 1] &= (uint8_t)(0xFF << (8 - tail));
        }
        {   // end magic and the stream's CRC (= the block's) behind the block's last bit
            uint64_t at = 32 + nbits;
            auto put = [&](uint64_t v, int count) { for (int k = count - 1; k >= 0; k--, at++) if ((v >> k) & 1) in[(size_t)(at >> 3)] |= (uint8_t)(0x80 >> (at & 7)); };
            put(MAGIC_END, 48); put(b.crc, 32);
            in.resize((size_t)((at + 7) >> 3));
        }
        BzStreamT z; memset(&z, 0, sizeof z);
        if (api.init(&z, 0, 0) != 0) return false;
        out.resize((size_t)b.level * 100000 + 4096);
        z.next_in = (char *)in.data(); z.avail_in = (unsigned)in.size();
        size_t got = 0;
        bool ok = false;
        for (;;) {
            z.next_out = (char *)out.data() + got; z.avail_out = (unsigned)(out.size() - got);
            const int rc = api.decompress(&z);
            got = out.size() - z.avail_out;
            if (rc == 4 /* BZ_STREAM_END */) { ok = true; break; }
            if (rc != 0) break;
            if (z.avail_out == 0) out.resize(out.size() * 2);        // (the first run-length layer may blow a block up beyond its nominal size)
            else if (z.avail_in == 0) break;                         // wants more input than the block has: not a block
        }
        api.end(&z);
        out.resize(ok ? got : 0);
        return ok;
    }

    void work()
    {
        for (;;) {
            size_t k;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return quit || (next_decode < blk_end && next_decode < limit_decode); });
                if (quit) return;
                k = next_decode++;
            }
            Slot &s = *slots[k];
            bool ok = false;
            try { ok = decode_block(blocks[k], s.out); } catch (const std::bad_alloc &) { ok = false; s.out.clear(); }
            { std::unique_lock<std::mutex> lk(mu); s.state = ok ? 1 : 2; cv_done.notify_all(); }
        }
    }

    // like a file read: < n only when the parallel part is over - then `handover` says where the one-stream decoder continues
    // (handover_byte: the first byte of the stream it starts at; handover_skip: bytes of that stream already delivered here)
    int read(uint8_t *dst, int n)
    {
        int got = 0;
        while (got < n && !handover) {
            if (next_consume >= blk_end) { handover = true; handover_byte = tail_byte; handover_skip = 0; handover_got_any = true; break; }
            Slot &s = *slots[next_consume];
            {
                std::unique_lock<std::mutex> lk(mu);
                if (limit_decode < next_consume + (size_t)nthreads * 3) { limit_decode = next_consume + (size_t)nthreads * 3; cv_work.notify_all(); }
                cv_done.wait(lk, [&] { return s.state != 0; });
            }
            const Block &b = blocks[next_consume];
            if (cur_off == 0 && next_consume == streams[b.stream].first_block) delivered_in_stream = 0;
            if (s.state == 2) {                                      // back to this stream's first byte with the one-stream decoder
                handover = true; handover_byte = streams[b.stream].byte0; handover_skip = delivered_in_stream; handover_got_any = b.stream > 0;
                break;
            }
            const size_t take = std::min(s.out.size() - cur_off, (size_t)(n - got));
            memcpy(dst + got, s.out.data() + cur_off, take);
            got += (int)take; cur_off += take; delivered_in_stream += take;
            if (cur_off == s.out.size()) { std::vector<uint8_t>().swap(s.out); next_consume++; cur_off = 0; }
        }
        return got;
    }
};

}   // namespace mcbz
