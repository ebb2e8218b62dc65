// mc_pgzip.h - parallel inflate of a .gz file (host side of the read sampler, mc_reader.cpp).
//
// open_file (reference microbe_census.py:47-59) hands gzip.open() to the parser: one inflate stream, ~0.5 GB/s of text, far below
// what the parser (and the GPU behind it) take.  A gzip member cannot be cut into independent pieces - every deflate block may
// refer back to the 32 KB in front of it - so the file is decoded SPECULATIVELY, the way pugz / rapidgzip do it:
//
//   * the compressed bytes are cut into chunks; a worker looks for a deflate block header behind its chunk's nominal start (a
//     dynamic-Huffman header that passes every consistency test of RFC 1951, or a gzip member header) and decodes from there with
//     an UNKNOWN window: a back-reference that reaches in front of the chunk becomes a marker symbol (16-bit output: 0..255 a byte,
//     256 + k = "byte k of the 32 KB in front of this chunk"), and copies of markers stay markers;
//   * the chunks are stitched in file order: chunk k counts only if chunk k - 1 ended exactly at the bit where k started (a false
//     block header is found out here at the latest: the chunk is then decoded again from the true position, sequentially, with
//     zlib and the known window); its markers are replaced from the last 32 KB of what lies in front of it, which is known by
//     then - that replacement and the CRC are again parallel over the chunks;
//   * chunk 0, and every repair, is plain zlib (raw inflate primed to the bit position, window set as dictionary, stopping at
//     block boundaries with Z_BLOCK).
// What comes out is byte for byte what zlib's gzread produces, for any gzip file: several members (bgzip, pigz -i, cat a.gz b.gz),
// stored and fixed-Huffman blocks, non-text data (decoded by the sequential repair path when the speculative one declines it);
// CRC-32 and ISIZE of every member are verified; a truncated or damaged file delivers what lies in front of the damage and then
// fails (gzip.open's EOFError / BadGzipFile), like the single-stream reader it replaces.
#pragma once
#include <new>
#include <zlib.h>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace mcgz {

// ------------------------------------------------------------------------------------------------------------------
// CRC-32 (gzip's, reflected 0xEDB88320) by carry-less multiplication: four 16-byte lanes folded per turn, then one, then the Barrett
// reduction (Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ Instruction", Intel 2009; the constants are
// x^(512+64), x^512, x^(128+64), x^128, x^64 mod P and floor(x^64 / P) in reflected form).  zlib 1.2.11's crc32 runs at 1 GB/s - a
// quarter of what a worker spent per chunk once the decoder itself got faster; this one at > 10 GB/s.  Chosen at run time (the CPU
// must have PCLMULQDQ and SSE4.1; checked once against zlib on a test pattern): crc32_any falls back to zlib's otherwise.
// ------------------------------------------------------------------------------------------------------------------
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("pclmul,sse4.1"))) inline uint32_t crc32_clmul(uint32_t crc0, const uint8_t *buf, size_t len)
{   // len >= 64 and a multiple of 16; crc0: the running CRC as zlib hands it out (pre- and post-inverted)
    const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596ll, 0x0154442bd4ll), k3k4 = _mm_set_epi64x(0x00ccaa009ell, 0x01751997d0ll);
    const __m128i k5k0 = _mm_set_epi64x(0, 0x0163cd6124ll), poly = _mm_set_epi64x(0x01f7011641ll, 0x01db710641ll);
    __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
    x1 = _mm_loadu_si128((const __m128i *)(buf + 0x00)); x2 = _mm_loadu_si128((const __m128i *)(buf + 0x10));
    x3 = _mm_loadu_si128((const __m128i *)(buf + 0x20)); x4 = _mm_loadu_si128((const __m128i *)(buf + 0x30));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)~crc0));
    x0 = k1k2;
    buf += 64; len -= 64;
    while (len >= 64) {
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x6 = _mm_clmulepi64_si128(x2, x0, 0x00); x7 = _mm_clmulepi64_si128(x3, x0, 0x00); x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x2 = _mm_clmulepi64_si128(x2, x0, 0x11); x3 = _mm_clmulepi64_si128(x3, x0, 0x11); x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
        y5 = _mm_loadu_si128((const __m128i *)(buf + 0x00)); y6 = _mm_loadu_si128((const __m128i *)(buf + 0x10));
        y7 = _mm_loadu_si128((const __m128i *)(buf + 0x20)); y8 = _mm_loadu_si128((const __m128i *)(buf + 0x30));
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5); x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
        x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7); x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
        buf += 64; len -= 64;
    }
    x0 = k3k4;                                                       // four lanes into one
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
    while (len >= 16) {
        x2 = _mm_loadu_si128((const __m128i *)buf);
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
        buf += 16; len -= 16;
    }
    x2 = _mm_clmulepi64_si128(x1, x0, 0x10);                         // 128 -> 64 bits
    x3 = _mm_setr_epi32(~0, 0, ~0, 0);
    x1 = _mm_srli_si128(x1, 8); x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_loadl_epi64(&k5k0);
    x2 = _mm_srli_si128(x1, 4); x1 = _mm_and_si128(x1, x3); x1 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_xor_si128(x1, x2);
    x0 = poly;                                                       // Barrett: 64 -> 32 bits
    x2 = _mm_and_si128(x1, x3); x2 = _mm_clmulepi64_si128(x2, x0, 0x10); x2 = _mm_and_si128(x2, x3); x2 = _mm_clmulepi64_si128(x2, x0, 0x00); x1 = _mm_xor_si128(x1, x2);
    return ~(uint32_t)_mm_extract_epi32(x1, 1);
}
inline bool crc32_clmul_usable()
{
    static const bool ok = [] {
        if (!__builtin_cpu_supports("pclmul") || !__builtin_cpu_supports("sse4.1")) return false;
        uint8_t t[64 * 5 + 48];
        for (size_t i = 0; i < sizeof t; i++) t[i] = (uint8_t)(i * 131u + (i >> 3) * 17u + 5u);
        return crc32_clmul(0x1234ABCDu, t, sizeof t) == (uint32_t)::crc32(0x1234ABCDu, t, (uInt)sizeof t) && crc32_clmul(0, t, 64) == (uint32_t)::crc32(0, t, 64);
    }();
    return ok;
}
#else
inline bool crc32_clmul_usable() { return false; }
inline uint32_t crc32_clmul(uint32_t c, const uint8_t *, size_t) { return c; }
#endif
// crc32() of zlib for any length: the bulk by crc32_clmul where the CPU has it
inline uint32_t crc32_any(uint32_t crc, const uint8_t *p, size_t n)
{
    if (n >= 256 && crc32_clmul_usable()) {
        const size_t bulk = n & ~(size_t)63;                              // (whole 64-byte blocks: every piece below - the last one too - is at least the 64 bytes crc32_clmul loads before it looks at its length)
        for (size_t o = 0; o < bulk;) { const size_t k = bulk - o > ((size_t)1 << 30) ? (size_t)1 << 30 : bulk - o; crc = crc32_clmul(crc, p + o, k); o += k; }
        p += bulk; n -= bulk;
    }
    while (n) { const size_t k = n > ((size_t)1 << 30) ? (size_t)1 << 30 : n; crc = (uint32_t)::crc32(crc, p, (uInt)k); p += k; n -= k; }
    return crc;
}

// ------------------------------------------------------------------------------------------------------------------
// bit reader over the mapped file (LSB first, as deflate packs its bits)
// ------------------------------------------------------------------------------------------------------------------
struct Bits {
    const uint8_t *base = nullptr, *p = nullptr, *end = nullptr;
    uint64_t buf = 0;
    int cnt = 0;                                                    // valid bits in buf
    void init(const uint8_t *b, const uint8_t *e, uint64_t bitpos)
    {
        base = b; end = e; p = b + (bitpos >> 3); buf = 0; cnt = 0;
        refill();
        const int skip = (int)(bitpos & 7);
        if (cnt >= skip) { buf >>= skip; cnt -= skip; } else cnt = -1;
    }
    inline void refill()
    {
        if (p + 8 <= end) {                                          // whole bytes that fit: one unaligned load
            uint64_t w;
            memcpy(&w, p, 8);
            buf |= w << cnt;
            const int take = (63 - cnt) >> 3;
            p += take; cnt += take * 8;
        } else {
            while (cnt <= 56 && p < end) { buf |= (uint64_t)*p++ << cnt; cnt += 8; }
        }
    }
    inline uint32_t peek(int n) const { return (uint32_t)(buf & ((1ull << n) - 1)); }
    inline void drop(int n) { buf >>= n; cnt -= n; }
    inline bool need(int n) { if (cnt < n) refill(); return cnt >= n; }
    uint64_t bitpos() const { return (uint64_t)(p - base) * 8 - (uint64_t)cnt; }
    void align_byte() { const int r = cnt & 7; buf >>= r; cnt -= r; }
};

// ------------------------------------------------------------------------------------------------------------------
// canonical Huffman decoding tables (primary table of PB bits; longer codes are walked bit by bit - they are rare)
// ------------------------------------------------------------------------------------------------------------------
template <int PB, int MAXSYM>
struct Huff {
    uint16_t tab[1 << PB];                                          // symbol << 4 | length (0: longer than PB bits, or no such code)
    uint16_t count[16], sorted[MAXSYM];                             // canonical form for the slow path
    int nsym = 0;
    // 0 ok, -1 over-subscribed, -2 incomplete (a single code of length 1 and the empty code are reported as ok: RFC 1951 allows them for distances)
    int build(const uint8_t *lens, int n)
    {
        nsym = n;
        memset(count, 0, sizeof count);
        for (int i = 0; i < n; i++) count[lens[i]]++;
        const int used = n - count[0];
        int left = 1;
        for (int len = 1; len <= 15; len++) { left <<= 1; left -= count[len]; if (left < 0) return -1; }
        uint16_t offs[16];
        offs[1] = 0;
        for (int len = 1; len < 15; len++) offs[len + 1] = (uint16_t)(offs[len] + count[len]);
        for (int i = 0; i < n; i++) if (lens[i]) sorted[offs[lens[i]]++] = (uint16_t)i;
        memset(tab, 0, sizeof tab);
        // codes in canonical order; the table is indexed by the bit-reversed code (the stream is read LSB first)
        uint32_t code = 0;
        int idx = 0;
        for (int len = 1; len <= 15; len++) {
            for (int k = 0; k < count[len]; k++, idx++, code++) {
                if (len > PB) continue;
                uint32_t rev = 0;
                for (int b = 0; b < len; b++) rev |= ((code >> b) & 1u) << (len - 1 - b);
                const uint16_t e = (uint16_t)((sorted[idx] << 4) | len);
                for (uint32_t j = rev; j < (1u << PB); j += 1u << len) tab[j] = e;
            }
            code <<= 1;
        }
        if (left > 0 && !(used <= 1)) return -2;
        return 0;
    }
    // returns the symbol, or -1 (no such code / not enough input)
    inline int decode(Bits &br) const
    {
        if (br.cnt < 15) br.refill();
        const uint16_t e = tab[br.peek(PB)];
        const int len = e & 15;
        if (len) { if (br.cnt < len) return -1; br.drop(len); return e >> 4; }
        // slow path (puff): walk the code bit by bit
        int code = 0, first = 0, index = 0;
        uint64_t b = br.buf;
        for (int l = 1; l <= 15; l++) {
            if (br.cnt < l) return -1;
            code |= (int)(b & 1); b >>= 1;
            const int c = count[l];
            if (code - c < first) { br.drop(l); return sorted[index + (code - first)]; }
            index += c; first += c; first <<= 1; code <<= 1;
        }
        return -1;
    }
};

typedef Huff<11, 288> LitHuff;
typedef Huff<9, 32> DistHuff;
typedef Huff<7, 19> ClHuff;

static const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
static const uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// Reads the header of a dynamic-Huffman block (the 3 header bits are already consumed) and builds its tables.  strict: refuse
// what zlib refuses AND what a real compressor never writes (used while guessing block starts).
inline bool is_text(int c) { return c < 0x80 && (c >= 0x20 || c == '\n' || c == '\r' || c == '\t'); }

inline bool read_dynamic(Bits &br, LitHuff &lit, DistHuff &dist, bool strict, bool *all_text = nullptr, uint8_t *lens_out = nullptr, int *hlit_out = nullptr, int *hdist_out = nullptr)
{
    if (!br.need(14)) return false;
    const int hlit = (int)br.peek(5) + 257; br.drop(5);
    const int hdist = (int)br.peek(5) + 1; br.drop(5);
    const int hclen = (int)br.peek(4) + 4; br.drop(4);
    if (hlit > 286 || hdist > 30) return false;
    uint8_t cl[19] = {0};
    for (int i = 0; i < hclen; i++) { if (!br.need(3)) return false; cl[kClOrder[i]] = (uint8_t)br.peek(3); br.drop(3); }
    ClHuff ch;
    if (ch.build(cl, 19) != 0) return false;                        // zlib: the code-length code must be complete
    { int used = 0; for (int i = 0; i < 19; i++) used += cl[i] != 0; if (used < 2) return false; }
    uint8_t lens[286 + 30 + 138];
    int n = 0;
    const int total = hlit + hdist;
    while (n < total) {
        const int sym = ch.decode(br);
        if (sym < 0) return false;
        if (sym < 16) lens[n++] = (uint8_t)sym;
        else {
            int rep, val = 0;
            if (sym == 16) { if (n == 0 || !br.need(2)) return false; val = lens[n - 1]; rep = 3 + (int)br.peek(2); br.drop(2); }
            else if (sym == 17) { if (!br.need(3)) return false; rep = 3 + (int)br.peek(3); br.drop(3); }
            else { if (!br.need(7)) return false; rep = 11 + (int)br.peek(7); br.drop(7); }
            if (n + rep > total) return false;
            while (rep--) lens[n++] = (uint8_t)val;
        }
    }
    if (lens[256] == 0) return false;                               // no end-of-block code
    if (lit.build(lens, hlit) != 0) return false;                    // (a lone literal/length code cannot be: 256 is one, data another)
    const int dr = dist.build(lens + hlit, hdist);
    if (dr != 0) return false;
    if (strict) { int used = 0; for (int i = 0; i < hlit; i++) used += lens[i] != 0; if (used < 3) return false; }
    if (all_text) { bool t = true; for (int i = 0; i < 256; i++) if (lens[i] && !is_text(i)) { t = false; break; } *all_text = t; }   // (a compressor gives codes to the bytes that occur)
    if (lens_out) { memcpy(lens_out, lens, (size_t)total); *hlit_out = hlit; *hdist_out = hdist; }
    return true;
}

// ------------------------------------------------------------------------------------------------------------------
// The tables the block loop of the speculative decoder runs on (round 5; the shape of libdeflate's): one 32-bit entry per code says
// everything the loop needs - what kind of symbol, how many bits the code takes, the literal or the BASE of the length / distance
// and how many extra bits follow - so that a symbol costs one table read and no second lookup (rounds 3 - 4: symbol table, then
// kLenBase / kLenExtra / kDistBase / kDistExtra, a refill test in front of every field).  Codes longer than the primary index go
// through a second-level table behind the primary one (the bit-by-bit walk of Huff::decode stays for the block HEADERS only).
//   entry: bits 0..7 bits to drop | 8..12 extra bits | 13..15 kind | 16..31 literal / base / offset of the second-level table
// TWO literals per lookup where both codes fit the primary index together (FT_LIT2: second literal in bits 24..31) - the bases of a
// FASTQ file have codes of two or three bits, and the chain lookup -> shift -> lookup is what a literal costs.
// ------------------------------------------------------------------------------------------------------------------
enum { FT_BAD = 0, FT_LIT = 1, FT_LEN = 2, FT_LIT2 = 3, FT_EOB = 4, FT_SUB = 6, FT_DIST = 2 };   // (the literal kinds are the odd ones)
#define MC_FT_KIND(e) (((e) >> 13) & 7u)
template <int PB, int SUBCAP>
struct FastTab {
    uint32_t t[(1 << PB) + SUBCAP];
    // lens[0 .. n): code lengths (a valid, complete or single-code set: Huff::build has seen them); dist: a distance code.  false:
    // the second-level tables do not fit (the caller declines the chunk; the sequential path decodes anything)
    bool build(const uint8_t *lens, int n, bool dist)
    {
        uint16_t count[16] = {0}, offs[16], sorted[288];
        for (int i = 0; i < n; i++) count[lens[i]]++;
        count[0] = 0;
        offs[1] = 0;
        for (int len = 1; len < 15; len++) offs[len + 1] = (uint16_t)(offs[len] + count[len]);
        for (int i = 0; i < n; i++) if (lens[i]) sorted[offs[lens[i]]++] = (uint16_t)i;
        memset(t, 0, sizeof(uint32_t) << PB);
        uint8_t subbits[1 << PB];
        bool anylong = false;
        for (int len = PB + 1; len <= 15; len++) if (count[len]) anylong = true;
        auto entry = [&](int sym) -> uint32_t {
            if (dist) return sym < 30 ? ((uint32_t)kDistBase[sym] << 16) | ((uint32_t)kDistExtra[sym] << 8) | ((uint32_t)FT_DIST << 13) : 0u;
            if (sym < 256) return ((uint32_t)sym << 16) | ((uint32_t)FT_LIT << 13);
            if (sym == 256) return (uint32_t)FT_EOB << 13;
            return sym - 257 < 29 ? ((uint32_t)kLenBase[sym - 257] << 16) | ((uint32_t)kLenExtra[sym - 257] << 8) | ((uint32_t)FT_LEN << 13) : 0u;
        };
        auto reverse = [](uint32_t code, int len) -> uint32_t {      // the low `len` bits of code, in reverse order
            uint32_t r = ((code & 0x5555u) << 1) | ((code >> 1) & 0x5555u);
            r = ((r & 0x3333u) << 2) | ((r >> 2) & 0x3333u); r = ((r & 0x0F0Fu) << 4) | ((r >> 4) & 0x0F0Fu); r = ((r & 0x00FFu) << 8) | (r >> 8);
            return r >> (16 - len);
        };
        uint32_t suboff[1 << PB];
        if (anylong) {                                               // how wide the second-level table of every primary prefix has to be
            memset(subbits, 0, sizeof subbits);
            uint32_t code = 0; int idx = 0;
            for (int len = 1; len <= 15; len++) {
                for (int k = 0; k < count[len]; k++, idx++, code++)
                    if (len > PB) { const uint32_t pre = reverse(code, len) & ((1u << PB) - 1); if (subbits[pre] < len - PB) subbits[pre] = (uint8_t)(len - PB); }
                code <<= 1;
            }
            uint32_t next = 1u << PB;
            for (uint32_t pre = 0; pre < (1u << PB); pre++)
                if (subbits[pre]) {
                    if (next + (1u << subbits[pre]) > (uint32_t)((1 << PB) + SUBCAP)) return false;
                    suboff[pre] = next;
                    memset(t + next, 0, sizeof(uint32_t) << subbits[pre]);
                    t[pre] = (next << 16) | ((uint32_t)subbits[pre] << 8) | ((uint32_t)FT_SUB << 13) | (uint32_t)PB;
                    next += 1u << subbits[pre];
                }
        }
        uint32_t code = 0; int idx = 0;
        for (int len = 1; len <= 15; len++) {
            for (int k = 0; k < count[len]; k++, idx++, code++) {
                const uint32_t rev = reverse(code, len), e = entry(sorted[idx]);
                if (len <= PB) { for (uint32_t j = rev; j < (1u << PB); j += 1u << len) t[j] = e | (uint32_t)len; }
                else {
                    const uint32_t pre = rev & ((1u << PB) - 1), sb = subbits[pre];
                    for (uint32_t j = rev >> PB; j < (1u << sb); j += 1u << (len - PB)) t[suboff[pre] + j] = e | (uint32_t)(len - PB);
                }
            }
            code <<= 1;
        }
        // pairs of literals: entry i = the literal at i followed by the literal at i >> its length, if that one is decided by the bits left
        // (from the top down: i >> l1 < i, so the entry looked at is still the single one)
        if (!dist)
            for (uint32_t i = (1u << PB) - 1; i > 0; i--) {
                const uint32_t e = t[i];
                if (MC_FT_KIND(e) != FT_LIT) continue;
                const uint32_t l1 = e & 0xFFu, e2 = t[i >> l1];
                if (MC_FT_KIND(e2) == FT_LIT && l1 + (e2 & 0xFFu) <= (uint32_t)PB) t[i] = (e & 0x00FF0000u) | ((e2 & 0x00FF0000u) << 8) | ((uint32_t)FT_LIT2 << 13) | (l1 + (e2 & 0xFFu));
            }
        return true;
    }
};
typedef FastTab<11, 2048> FastLit;
typedef FastTab<9, 1024> FastDist;

inline void fixed_tables(LitHuff &lit, DistHuff &dist)
{
    uint8_t l[288], d[30];
    for (int i = 0; i < 144; i++) l[i] = 8;
    for (int i = 144; i < 256; i++) l[i] = 9;
    for (int i = 256; i < 280; i++) l[i] = 7;
    for (int i = 280; i < 288; i++) l[i] = 8;
    for (int i = 0; i < 30; i++) d[i] = 5;
    lit.build(l, 288); dist.build(d, 30);
}

// gzip member header at p (RFC 1952); returns its length, 0 if p does not hold one, -1 if the file ends inside it
inline long member_header(const uint8_t *p, const uint8_t *end)
{
    if (end - p < 10) return (end - p >= 2 && p[0] == 0x1f && p[1] == 0x8b) || end - p < 2 ? -1 : 0;
    if (p[0] != 0x1f || p[1] != 0x8b) return 0;
    if (p[2] != 8 || (p[3] & 0xE0)) return 0;
    const int flg = p[3];
    const uint8_t *q = p + 10;
    if (flg & 4) { if (end - q < 2) return -1; const size_t xl = q[0] | (q[1] << 8); q += 2; if ((size_t)(end - q) < xl) return -1; q += xl; }
    if (flg & 8) { while (q < end && *q) q++; if (q >= end) return -1; q++; }
    if (flg & 16) { while (q < end && *q) q++; if (q >= end) return -1; q++; }
    if (flg & 2) { if (end - q < 2) return -1; q += 2; }
    return (long)(q - p);
}

// ------------------------------------------------------------------------------------------------------------------
// what a decoded chunk consists of
// ------------------------------------------------------------------------------------------------------------------
struct MemberEnd { uint64_t out_pos; uint32_t crc, isize; };      // a member ended after out_pos bytes of this chunk's output

struct SymBuf {                                                    // 16-bit symbols; grows without touching what it has not written
    uint16_t *p = nullptr; size_t n = 0, cap = 0;
    ~SymBuf() { free(p); }
    SymBuf() {}
    SymBuf(const SymBuf &) = delete;
    SymBuf &operator=(const SymBuf &) = delete;
    void reserve(size_t c, size_t keep = 0) { if (c <= cap) return; uint16_t *q = (uint16_t *)malloc(c * 2); if (keep) memcpy(q, p, keep * 2); free(p); p = q; cap = c; }
    uint16_t *data() { return p; }
    size_t size() const { return n; }
    void drop() { free(p); p = nullptr; n = cap = 0; }
    void swap(SymBuf &o) { std::swap(p, o.p); std::swap(n, o.n); std::swap(cap, o.cap); }
};

// The bytes of a chunk: like std::vector<uint8_t> for what this file does with it, but resize() never writes - a vector zero-fills
// what it grows by, 5 MB per chunk on the ONE thread that stitches the chunks (the reader's; round 5: that thread, not the
// workers, bounded the sampling rate of a .gz file beyond 8 workers).
struct ByteBuf {
    uint8_t *p = nullptr; size_t n = 0, cap = 0;
    ByteBuf() {}
    ~ByteBuf() { free(p); }
    ByteBuf(const ByteBuf &) = delete;
    ByteBuf &operator=(const ByteBuf &) = delete;
    ByteBuf(ByteBuf &&o) noexcept : p(o.p), n(o.n), cap(o.cap) { o.p = nullptr; o.n = o.cap = 0; }
    uint8_t *data() { return p; }
    const uint8_t *data() const { return p; }
    size_t size() const { return n; }
    size_t capacity() const { return cap; }
    void clear() { n = 0; }
    void resize(size_t m)                                            // (what lay in [0, min(n, m)) stays)
    {
        if (m > cap) {
            uint8_t *q = (uint8_t *)malloc(m ? m : 1);
            if (!q) throw std::bad_alloc();                            // (as the std::vector this replaced: the workers mark their chunk, the reader reports it)
            if (n) memcpy(q, p, n);
            free(p); p = q; cap = m;
        }
        n = m;
    }
    void swap(ByteBuf &o) { std::swap(p, o.p); std::swap(n, o.n); std::swap(cap, o.cap); }
    void drop() { free(p); p = nullptr; n = cap = 0; }
    const uint8_t *begin() const { return p; }
    const uint8_t *end() const { return p + n; }
};

// Buffers that outlive one reader: the slices of a file (ParallelGz::start_slice) are read by one reader each, one after the other, and
// every slice decodes ALL its chunks at once - 33 chunks x 16 MB of fresh pages per slice are 135,000 page faults from a dozen threads
// that queue up in the kernel.  A reader hands what it holds to the pool when it closes; the pool lives as long as a reader holds it
// (the next slice's reader is opened before this one's is closed).
struct SharedBufs {
    enum { MAX = 160 };
    std::mutex mu;
    std::vector<std::unique_ptr<SymBuf>> sym;
    std::vector<ByteBuf> bytes;
    ~SharedBufs()
    {   // (the last reader of a run closes: 160 buffers of 16 MB are 100 ms of munmap on the thread the caller waits for - not there)
        if (sym.empty() && bytes.empty()) return;
        auto *a = new std::vector<std::unique_ptr<SymBuf>>(std::move(sym));
        auto *b = new std::vector<ByteBuf>(std::move(bytes));
        try { std::thread([a, b] { delete a; delete b; }).detach(); } catch (...) { delete a; delete b; }
    }
    static std::shared_ptr<SharedBufs> get()
    {
        static std::mutex m; static std::weak_ptr<SharedBufs> w;
        std::unique_lock<std::mutex> lk(m);
        std::shared_ptr<SharedBufs> s = w.lock();
        if (!s) { s = std::make_shared<SharedBufs>(); w = s; }
        return s;
    }
    void give(SymBuf &b) { if (!b.cap) return; std::unique_lock<std::mutex> lk(mu); if (sym.size() < MAX) { sym.emplace_back(new SymBuf()); sym.back()->swap(b); b.n = 0; sym.back()->n = 0; return; } lk.unlock(); b.drop(); }
    void give(ByteBuf &b) { if (!b.capacity()) return; std::unique_lock<std::mutex> lk(mu); if (bytes.size() < MAX) { bytes.emplace_back(); bytes.back().swap(b); bytes.back().clear(); return; } lk.unlock(); b.drop(); }
    bool take(SymBuf &b) { std::unique_lock<std::mutex> lk(mu); if (sym.empty()) return false; b.swap(*sym.back()); sym.pop_back(); b.n = 0; return true; }
    bool take(ByteBuf &b) { std::unique_lock<std::mutex> lk(mu); if (bytes.empty()) return false; b.swap(bytes.back()); bytes.pop_back(); b.clear(); return true; }
};

struct Chunk {
    uint64_t nominal_bit = 0;                                       // where the search for its first block started
    uint64_t start_bit = 0, end_bit = 0;                            // first block decoded / first bit behind the last one
    bool found = false;                                             // a start was found and decoded to the chunk's end
    bool starts_member = false;                                     // start_bit is the first deflate bit of a gzip member (window empty: no markers)
    bool at_eof = false;                                            // the data ended with this chunk (last member complete)
    bool bad = false; std::string msg;                              // damage met while decoding from a KNOWN position
    SymBuf sym;                                                     // speculative output (markers): the first sym.size() symbols of the chunk
    size_t total = 0;                                               // ... of `total`: what lies behind them was decoded straight into bytes (Spec::run)
    ByteBuf bytes;                                                  // final output
    std::vector<MemberEnd> ends;
    std::vector<uint32_t> seg_crc;                                  // crc of bytes between member ends (ends.size() + 1 segments)
    bool have_bytes = false, skipped = false;                      // skipped: the chunk in front of it covered it entirely
    // scheduling
    int state = 0;                                                  // 0 queued, 1 decoded, 2 resolved
};

// ------------------------------------------------------------------------------------------------------------------
// speculative decoding of one chunk
// ------------------------------------------------------------------------------------------------------------------
struct Spec {
    const uint8_t *base, *end;                                      // the whole file
    uint64_t data_end_bit;
    LitHuff lit; DistHuff dist;
    FastLit flit; FastDist fdist;                                   // the tables the block loop runs on (built per block from the same code lengths)
    bool fixed_ready = false; FastLit fixed_lit; FastDist fixed_dist;
    bool allow_plain = true;                                        // (tools/pgz_bench.cpp measures the symbol form alone with this off)

    // One Huffman block, from behind its header to its end-of-block code.  The bit buffer is refilled ONCE per turn to at least 56
    // bits - a literal / length code (15), its extra bits (5), a distance code (15) and its extra bits (13) are 48 - so nothing in a
    // turn tests for input; up to three literals leave per refill.  Near the end of the file the refill takes what is left and the
    // turn checks afterwards that it did not consume more than there was (cnt < 0).  Matches are copied 16 bytes at a time
    // (overlapping allowed from that distance on; the buffer has 320 symbols of slack behind pos).
    // OT = uint16_t: symbols with markers (the window in front of the chunk is unknown; a back-reference may reach up to 32 KB in
    // front of the chunk); OT = uint8_t: plain bytes - no reference may reach below `floor` (the start of the member, or of the
    // marker-free 32 KB the caller switched on).  grow(pos): more room behind pos, returns the (new) base.
    // CHECK_LIT: a fixed-Huffman block met with an unknown window has a code for every byte - every literal is looked at.
    template <class OT, bool CHECK_LIT, class GROW>
    inline bool block(Bits &br, OT *&o, size_t &pos, size_t &cap, long floor, GROW &&grow)
    {
        const uint32_t *LT = flit.t, *DT = fdist.t;
        uint64_t buf = br.buf;
        int cnt = br.cnt;
        const uint8_t *p = br.p, *const end8 = br.end - 8;
        constexpr bool MARKERS = sizeof(OT) == 2;
        constexpr long WORD = 16 / (long)sizeof(OT);                 // symbols per 16-byte move
        bool ok = false;
        for (;;) {
            if (pos + 320 > cap) o = grow(pos, cap);
#define MC_GZ_REFILL()                                                                                                   \
    do {                                                                                                                 \
        if (p <= end8) { uint64_t w_; memcpy(&w_, p, 8); buf |= w_ << cnt; p += (63 - cnt) >> 3; cnt |= 56; }            \
        else { while (cnt <= 56 && p < br.end) { buf |= (uint64_t)*p++ << cnt; cnt += 8; } }                             \
    } while (0)
            MC_GZ_REFILL();
            uint32_t e = LT[buf & 0x7FFu];
            if (e & 0x2000u) {                                       // literals: up to three lookups per refill (each takes at most 11 bits), one or two literals per lookup
#define MC_GZ_PUT()                                                                                                               \
    do {                                                                                                                          \
        if (CHECK_LIT && (!is_text((int)((e >> 16) & 0xFFu)) || ((e & 0x4000u) && !is_text((int)(e >> 24))))) goto done;             \
        if (MARKERS) { const uint32_t two_ = ((e >> 16) & 0xFFu) | ((e >> 24) << 16); memcpy(o + pos, &two_, 4); }                   \
        else { const uint16_t two_ = (uint16_t)(e >> 16); memcpy(o + pos, &two_, 2); }                                               \
        pos += 1u + ((e >> 14) & 1u);                                                                                             \
        buf >>= (e & 0xFFu); cnt -= (int)(e & 0xFFu);                                                                             \
    } while (0)
                MC_GZ_PUT();
                e = LT[buf & 0x7FFu];
                if (e & 0x2000u) {
                    MC_GZ_PUT();
                    e = LT[buf & 0x7FFu];
                    if (e & 0x2000u) {
                        MC_GZ_PUT();
                        if (cnt < 0) break;
                        continue;
                    }
                }
                if (cnt < 0) break;
                MC_GZ_REFILL();                                      // the symbol in e has not been consumed: the same bits are still at the bottom
            }
            if (MC_FT_KIND(e) == FT_SUB) {                           // a code longer than 11 bits: the second-level table
                buf >>= 11; cnt -= 11;
                e = LT[(e >> 16) + (uint32_t)(buf & ((1u << ((e >> 8) & 31u)) - 1u))];
            }
            buf >>= (e & 0xFFu); cnt -= (int)(e & 0xFFu);
            const uint32_t kind = MC_FT_KIND(e);
            if (kind == FT_LIT) { if (CHECK_LIT && !is_text((int)(e >> 16))) break; o[pos++] = (OT)(e >> 16); if (cnt < 0) break; continue; }   // (from a second-level table: single)
            if (kind == FT_EOB) { ok = cnt >= 0; break; }
            if (kind != FT_LEN) break;                               // no such code
            const int xl = (int)((e >> 8) & 31u);
            const int len = (int)(e >> 16) + (int)(buf & ((1u << xl) - 1u));
            buf >>= xl; cnt -= xl;
            uint32_t de = DT[buf & 0x1FFu];
            if (MC_FT_KIND(de) == FT_SUB) { buf >>= 9; cnt -= 9; de = DT[(de >> 16) + (uint32_t)(buf & ((1u << ((de >> 8) & 31u)) - 1u))]; }
            if (MC_FT_KIND(de) != FT_DIST) break;
            buf >>= (de & 0xFFu); cnt -= (int)(de & 0xFFu);
            const int xd = (int)((de >> 8) & 31u);
            const long d = (long)(de >> 16) + (long)(buf & ((1u << xd) - 1u));
            buf >>= xd; cnt -= xd;
            if (cnt < 0) break;
            long src = (long)pos - d;
            if (src < floor) break;                                  // (markers: floor = -32768)
            if (!MARKERS || src >= 0) {
                OT *dst = o + pos;
                const OT *sp = o + src;
                if (d >= WORD) { OT *const de_ = dst + len; do { memcpy(dst, sp, 16); dst += WORD; sp += WORD; } while (dst < de_); }
                else if (!MARKERS && d >= 8) { OT *const de_ = dst + len; do { memcpy(dst, sp, 8); dst += 8; sp += 8; } while (dst < de_); }
                else for (int i = 0; i < len; i++) dst[i] = sp[i];
                pos += (size_t)len;
            } else {
                for (int i = 0; i < len; i++, src++) o[pos++] = src < 0 ? (OT)(256 + 32768 + src) : o[src];
            }
        }
    done:
#undef MC_GZ_REFILL
#undef MC_GZ_PUT
        br.buf = buf; br.cnt = cnt; br.p = p;
        return ok;
    }

    // Decodes blocks from `bit` until a block starts at or behind stop_bit (or the data ends).  Returns false on anything that cannot
    // be deflate data (or that a text file would not hold, while the window is unknown).
    // The output starts as 16-bit symbols with markers (c.sym) - the window in front of the chunk is unknown.  It goes on as PLAIN
    // BYTES, written straight to their final place in c.bytes, as soon as nothing behind can be a marker any more (round 5; rapidgzip
    // does the same): when a member starts inside the chunk (or the chunk starts with one: a bgzip file never sees a marker), or
    // when, at a block boundary, the last 32 KB hold no marker - deflate cannot reach further back.  On FASTQ text that is the case
    // a few hundred KB into a chunk: the rest is decoded at half the memory traffic and needs no marker replacement.
    // c.sym.size() = the symbols in front of the switch, c.total = the whole output.
    bool run(uint64_t bit, uint64_t stop_bit, bool member_start, Chunk &c)
    {
        SymBuf &out = c.sym;
        c.ends.clear();
        size_t cap = out.cap < (1u << 20) ? (1u << 20) : out.cap;
        out.reserve(cap);
        uint16_t *o = out.data();
        size_t pos = 0;
        bool plain = false;                                          // the output has gone over to bytes
        size_t nsym = 0, bcap = 0, next_scan = 32768;                // symbols in front of the switch; capacity of c.bytes; where the next look for markers pays
        uint8_t *ob = nullptr;
        long floor = -32768;                                         // lowest index a back-reference may reach
        auto grow16 = [&](size_t at, size_t &cp) -> uint16_t * { cp *= 2; out.reserve(cp, at); return out.data(); };
        auto grow8 = [&](size_t at, size_t &cp) -> uint8_t * { cp *= 2; c.bytes.resize(at); c.bytes.resize(cp); return c.bytes.data(); };   // (only the bytes in use are copied)
        auto to_plain = [&](size_t keep_from) {                      // bytes [keep_from, pos) are marker-free and may be referenced from now on
            nsym = pos;
            bcap = std::max<size_t>(std::max<size_t>(c.bytes.capacity(), cap), pos + (1u << 20));
            c.bytes.resize(bcap);
            ob = c.bytes.data();
            for (size_t k = keep_from; k < pos; k++) ob[k] = (uint8_t)o[k];
            plain = true; floor = (long)keep_from;
        };
        Bits br;
        br.init(base, end, bit);
        if (br.cnt < 0) return false;
        if (member_start) to_plain(0);
        for (;;) {
            const uint64_t here = br.bitpos();
            if (here >= stop_bit && pos > 0) { c.end_bit = here; break; }
            if (!plain && allow_plain && pos >= next_scan) {                        // a marker in the last 32 KB?  (looked for from the end: the latest one says when to look again)
                size_t k = pos;
                const size_t lo = pos - 32768;
                while (k > lo) {
                    if (k - lo >= 8) { uint64_t a, b; memcpy(&a, o + k - 8, 8); memcpy(&b, o + k - 4, 8); if (((a | b) & 0xFF00FF00FF00FF00ull) == 0) { k -= 8; continue; } }
                    if (o[k - 1] >= 256) break;
                    k--;
                }
                if (k == lo) to_plain(lo); else next_scan = k + 32768;
            }
            if (!br.need(3)) return false;
            const int bfinal = (int)br.peek(1), btype = (int)(br.peek(3) >> 1);
            br.drop(3);
            if (btype == 3) return false;
            if (btype == 0) {
                br.align_byte();
                if (!br.need(32)) return false;
                const uint32_t len = br.peek(16); br.drop(16);
                const uint32_t nlen = br.peek(16); br.drop(16);
                if ((len ^ 0xFFFFu) != nlen) return false;
                // the bytes follow byte-aligned: take them from memory
                const uint8_t *src = base + (br.bitpos() >> 3);
                if ((uint64_t)(end - src) < len) return false;
                if (plain) {
                    if (pos + len + 320 > bcap) { bcap = std::max(bcap * 2, pos + len + 65536); c.bytes.resize(bcap); ob = c.bytes.data(); }
                    memcpy(ob + pos, src, len);
                } else {
                    if (pos + len + 320 > cap) { cap = std::max(cap * 2, pos + len + 65536); out.reserve(cap, pos); o = out.data(); }
                    for (uint32_t i = 0; i < len; i++) o[pos + i] = src[i];
                }
                pos += len;
                br.init(base, end, (uint64_t)(src + len - base) * 8);
            } else {
                bool all_text = false;
                const bool check_lit = !plain && btype == 1;               // (a fixed-Huffman block has a code for every byte: look at each literal)
                if (btype == 1) {
                    if (!fixed_ready) { uint8_t l[288], d[30]; for (int i = 0; i < 144; i++) l[i] = 8; for (int i = 144; i < 256; i++) l[i] = 9; for (int i = 256; i < 280; i++) l[i] = 7; for (int i = 280; i < 288; i++) l[i] = 8; for (int i = 0; i < 30; i++) d[i] = 5; if (!fixed_lit.build(l, 288, false) || !fixed_dist.build(d, 30, true)) return false; fixed_ready = true; }
                    memcpy(flit.t, fixed_lit.t, sizeof(uint32_t) << 11); memcpy(fdist.t, fixed_dist.t, sizeof(uint32_t) << 9);   // (all codes of the fixed block fit the primary tables)
                } else {
                    uint8_t lens[286 + 30 + 138];
                    int hlit = 0, hdist = 0;
                    if (!read_dynamic(br, lit, dist, false, &all_text, lens, &hlit, &hdist)) return false;
                    if (!plain && !all_text) return false;                  // not text: left to the sequential path
                    if (!flit.build(lens, hlit, false) || !fdist.build(lens + hlit, hdist, true)) return false;
                }
                const bool okb = plain ? block<uint8_t, false>(br, ob, pos, bcap, floor, grow8)
                                       : check_lit ? block<uint16_t, true>(br, o, pos, cap, floor, grow16) : block<uint16_t, false>(br, o, pos, cap, floor, grow16);
                if (!okb) return false;
            }
            if (bfinal) {
                // end of a member: trailer, then either the end of the data or the next member
                br.align_byte();
                const uint8_t *t = base + (br.bitpos() >> 3);
                if (end - t < 8) return false;
                MemberEnd me; me.out_pos = pos;
                me.crc = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
                me.isize = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
                c.ends.push_back(me);
                t += 8;
                while (t < end && *t == 0) t++;                          // (zero padding between / behind members)
                if (t >= end) { c.end_bit = (uint64_t)(end - base) * 8; c.at_eof = true; break; }                // end of the data
                const long h = member_header(t, end);
                if (h <= 0) return false;                                // (bytes that are no member: gzip.open raises BadGzipFile there - the sequential path reports it)
                br.init(base, end, (uint64_t)(t + h - base) * 8);
                if (!plain) to_plain(pos); else floor = (long)pos;       // a new member: nothing in front of it can be referenced
            }
        }
        out.n = plain ? nsym : pos;
        c.total = pos;
        return true;
    }

    // A bit position at or behind `from` (and in front of `limit`) where a block may start, verified by decoding the chunk from it.
    bool find_and_decode(uint64_t from, uint64_t limit, uint64_t stop_bit, Chunk &c)
    {
        const uint64_t last = std::min<uint64_t>(limit, data_end_bit > 192 ? data_end_bit - 192 : 0);
        for (uint64_t bit = from; bit < last; bit++) {
            const uint8_t *p = base + (bit >> 3);
            if ((bit & 7) == 0 && p[0] == 0x1f && p[1] == 0x8b && p[2] == 8 && !(p[3] & 0xE0)) {     // a gzip member header (bgzip, pigz -i, concatenated files)
                const long h = member_header(p, end);
                if (h > 0 && run((uint64_t)(p + h - base) * 8, stop_bit, true, c)) { c.start_bit = (uint64_t)(p + h - base) * 8; c.starts_member = true; return true; }
            }
            uint64_t lo, hi;
            memcpy(&lo, p, 8); memcpy(&hi, p + 8, 8);
            const int sh = (int)(bit & 7);
            const uint64_t w = sh ? (lo >> sh) | (hi << (64 - sh)) : lo;      // 64 bits of the stream from `bit`
            if ((w & 6u) != 4u) continue;                                    // BTYPE = 2 (a member's last block - BFINAL = 1 - starts many chunks of a bgzip file)
            {   // the cheap tests of read_dynamic first: counts in range, code-length code complete (most positions end here)
                if (((w >> 3) & 31u) > 29u || ((w >> 8) & 31u) > 29u) continue;
                const int hclen = (int)((w >> 13) & 15u) + 4;
                const uint64_t w2 = sh ? (hi >> sh) : hi;                    // bits 64 .. of the stream (at least 56 of them)
                int kraft = 0, used = 0;
                for (int i = 0; i < hclen; i++) {
                    const int at = 17 + 3 * i;
                    const uint32_t l = at + 3 <= 64 ? (uint32_t)(w >> at) & 7u : at >= 64 ? (uint32_t)(w2 >> (at - 64)) & 7u : (uint32_t)((w >> at) | (w2 << (64 - at))) & 7u;
                    if (l) { kraft += 128 >> l; used++; }
                }
                if (kraft != 128 || used < 2) continue;
            }
            Bits br;
            br.init(base, end, bit + 3);
            if (!read_dynamic(br, lit, dist, true)) continue;
            if (run(bit, stop_bit, false, c)) { c.start_bit = bit; c.starts_member = false; return true; }
        }
        return false;
    }
};

// ------------------------------------------------------------------------------------------------------------------
// sequential decoding from a known position with a known window (zlib): chunk 0 and repairs
// ------------------------------------------------------------------------------------------------------------------
// Decodes from `bit` (inside a member whose last 32 KB are `window`, window_n bytes; or at the first deflate bit of a member when
// member_start) until a block boundary at or behind stop_bit, the end of the data, or damage.  Fills c.bytes / c.ends / c.end_bit.
inline void decode_known(const uint8_t *base, const uint8_t *end, uint64_t bit, uint64_t stop_bit, const uint8_t *window, size_t window_n, Chunk &c)
{
    c.bytes.clear(); c.ends.clear(); c.bad = false; c.at_eof = false;
    ByteBuf &out = c.bytes;
    size_t cap = std::max<size_t>(out.capacity(), 1u << 20);
    out.resize(cap);
    size_t pos = 0;
    for (;;) {                                                       // one member (or the rest of one) per turn
        z_stream z;
        memset(&z, 0, sizeof z);
        if (inflateInit2(&z, -15) != Z_OK) { c.bad = true; c.msg = "inflateInit2 failed"; break; }
        const uint8_t *p = base + (bit >> 3);
        const int skip = (int)(bit & 7);
        if (skip) { inflatePrime(&z, 8 - skip, *p >> skip); p++; }
        if (window_n) inflateSetDictionary(&z, window, (uInt)window_n);
        z.next_in = (Bytef *)p;
        const uint8_t *in0 = p;
        bool member_done = false, stop = false;
        for (;;) {
            if (cap - pos < (1u << 16)) { cap *= 2; out.resize(cap); }
            z.avail_in = (uInt)std::min<size_t>((size_t)(end - z.next_in), 1u << 30);
            z.next_out = out.data() + pos;
            z.avail_out = (uInt)std::min<size_t>(cap - pos, 1u << 30);
            const int rc = inflate(&z, Z_BLOCK);
            pos = (size_t)(z.next_out - out.data());
            if (rc == Z_STREAM_END) { member_done = true; break; }
            if (rc != Z_OK && rc != Z_BUF_ERROR) { c.bad = true; c.msg = z.msg ? z.msg : "invalid deflate data"; break; }
            if (rc == Z_BUF_ERROR && z.avail_in == 0 && z.next_in >= end && z.avail_out) { c.bad = true; c.msg = "compressed file ended before the end-of-stream marker was reached"; break; }
            if ((z.data_type & 128) && !(z.data_type & 64)) {        // at a block boundary (not the end of the last block)
                const uint64_t here = ((uint64_t)(z.next_in - base) * 8) - (uint64_t)(z.data_type & 63);
                if (here >= stop_bit && pos > 0) { c.end_bit = here; stop = true; break; }
            }
        }
        const uint8_t *t = z.next_in;                                // (after Z_STREAM_END: the byte behind the last deflate byte)
        (void)in0;
        inflateEnd(&z);
        if (c.bad || stop) break;
        if (member_done) {
            if (end - t < 8) { c.bad = true; c.msg = "compressed file ended before the end-of-stream marker was reached"; break; }
            MemberEnd me; me.out_pos = pos;
            me.crc = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
            me.isize = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
            c.ends.push_back(me);
            t += 8;
            while (t < end && *t == 0) t++;
            if (t >= end) { c.end_bit = (uint64_t)(end - base) * 8; c.at_eof = true; break; }
            const long h = member_header(t, end);
            // non-zero bytes behind a member that are no member header: gzip.open raises BadGzipFile("Not a gzipped file") when it gets
            // there (GzipFile._read_gzip_header), after the zero padding _read_eof skips - what was decoded so far is delivered first
            if (h == 0) { c.bad = true; c.msg = "Not a gzipped file: bytes behind the last member"; c.end_bit = (uint64_t)(end - base) * 8; break; }
            if (h < 0) { c.bad = true; c.msg = "compressed file ended before the end-of-stream marker was reached"; break; }
            bit = (uint64_t)(t + h - base) * 8;
            window_n = 0;
            if (bit >= stop_bit) { c.end_bit = bit; c.starts_member = false; break; }   // the next chunk starts with this member (its worker found the same header)
        }
    }
    out.resize(pos);
    c.have_bytes = true;
}

// ------------------------------------------------------------------------------------------------------------------
// One inflate stream over the mapped file, with gzip.open's rules for what lies between and behind the members (GzipFile._read_eof /
// _read_gzip_header): zero padding is skipped, another member goes on, anything else is BadGzipFile - zlib's gzread stops at zero
// padding and ignores other bytes.  Used where one thread is allowed (-t 1) and for the quality-offset peek; ParallelGz follows
// the same rules.
// ------------------------------------------------------------------------------------------------------------------
class SerialGz {
public:
    SerialGz(const uint8_t *data, size_t n) : p_(data), end_(data + n) { memset(&z_, 0, sizeof z_); }
    ~SerialGz() { if (init_) inflateEnd(&z_); }
    bool start()
    {
        const long h = member_header(p_, end_);
        if (h <= 0 || inflateInit2(&z_, -15) != Z_OK) return false;
        init_ = true; p_ += h; in_member_ = true;
        return true;
    }
    // like a file read: < n only at the end of the data; bad: the file is truncated / damaged behind what was delivered
    int read(uint8_t *dst, int n, bool *bad, std::string *msg)
    {
        int got = 0;
        *bad = false;
        while (got < n && !done_) {
            if (!in_member_) {
                while (p_ < end_ && *p_ == 0) p_++;
                if (p_ >= end_) { done_ = true; break; }
                const long h = member_header(p_, end_);
                if (h == 0) { fail("Not a gzipped file: bytes behind the last member"); break; }
                if (h < 0) { fail("compressed file ended before the end-of-stream marker was reached"); break; }
                p_ += h;
                inflateReset(&z_);
                crc_ = 0; isize_ = 0; in_member_ = true;
            }
            z_.next_in = (Bytef *)p_;
            z_.avail_in = (uInt)std::min<size_t>((size_t)(end_ - p_), 1u << 30);
            z_.next_out = dst + got;
            z_.avail_out = (uInt)(n - got);
            const int rc = inflate(&z_, Z_NO_FLUSH);
            const int made = (int)(z_.next_out - (dst + got));
            crc_ = (uint32_t)crc32(crc_, dst + got, (uInt)made);
            isize_ += (uint32_t)made;
            got += made;
            p_ = z_.next_in;
            if (rc == Z_STREAM_END) {
                if (end_ - p_ < 8) { fail("compressed file ended before the end-of-stream marker was reached"); break; }
                const uint32_t crc = (uint32_t)p_[0] | ((uint32_t)p_[1] << 8) | ((uint32_t)p_[2] << 16) | ((uint32_t)p_[3] << 24);
                const uint32_t isz = (uint32_t)p_[4] | ((uint32_t)p_[5] << 8) | ((uint32_t)p_[6] << 16) | ((uint32_t)p_[7] << 24);
                if (crc != crc_) { fail("CRC check failed"); break; }
                if (isz != isize_) { fail("Incorrect length of data produced"); break; }
                p_ += 8; in_member_ = false;
            } else if (rc != Z_OK && rc != Z_BUF_ERROR) { fail(z_.msg ? z_.msg : "invalid deflate data"); break; }
            else if (made == 0 && p_ >= end_) { fail("compressed file ended before the end-of-stream marker was reached"); break; }
        }
        if (failed_) { *bad = true; *msg = err_; }
        return got;
    }
private:
    void fail(const char *m) { failed_ = true; done_ = true; err_ = m; }
    const uint8_t *p_, *end_;
    z_stream z_;
    bool init_ = false, in_member_ = false, done_ = false, failed_ = false;
    uint32_t crc_ = 0, isize_ = 0;
    std::string err_;
};

// ------------------------------------------------------------------------------------------------------------------
// the reader: read() delivers the decompressed bytes in order
// ------------------------------------------------------------------------------------------------------------------
class ParallelGz {
public:
    // data: the mapped file [data, data + n); threads >= 2
    ParallelGz(const uint8_t *data, size_t n, int threads, size_t chunk_bytes = (size_t)1 << 20)
        : base_(data), end_(data + n), nthreads_(threads), chunk_bytes_(chunk_bytes) {}
    ~ParallelGz()
    {
        stop();
        if (shared_) {                                                  // (a slice: what this reader holds goes to the next one)
            for (auto &c : chunks_) { shared_->give(c->sym); shared_->give(c->bytes); }
            for (auto &b : pool_sym_) shared_->give(*b);
            for (auto &b : pool_bytes_) shared_->give(b);
        }
        if (getenv("MC_PGZ_DEBUG")) fprintf(stderr, "pgzip: %zu chunks: %zu speculative, %zu sequential (%zu of them had found no start), %zu skipped; %.1f %% of the speculative output decoded as plain bytes; worker seconds: decode %.3f, markers %.3f, crc %.3f; consumer: sequential decode %.3f, copies to the reader %.3f, waiting for chunks %.3f\n",
                                            chunks_.size(), n_spec_, n_seq_, n_notfound_, n_skip_, 100.0 * (double)b_plain_ / (double)(b_spec_ ? b_spec_ : 1), t_decode_.load() * 1e-9, t_resolve_.load() * 1e-9, t_crc_.load() * 1e-9, t_seq_ * 1e-9, t_copy_ * 1e-9, t_wait_ * 1e-9);
    }

    // false: not a gzip file this reader handles (the caller falls back to gzread)
    bool start()
    {
        const long h = member_header(base_, end_);
        if (h <= 0) return false;
        first_bit_ = (uint64_t)h * 8;
        const uint64_t data_bits = (uint64_t)(end_ - base_) * 8;
        // chunk k looks for its start from bit first_bit_ + k * chunk_bytes_ * 8
        for (uint64_t b = first_bit_; b < data_bits; b += (uint64_t)chunk_bytes_ * 8) {
            std::unique_ptr<Chunk> c(new Chunk());
            c->nominal_bit = b;
            chunks_.push_back(std::move(c));
        }
        lim_ = chunks_.size();
        for (int i = 0; i < nthreads_; i++) workers_.emplace_back([this] { work(); });
        return true;
    }

    // ---- a SLICE of the file for one rank of a multi-GPU run (microbecensus_amd/distributed.py): chunks [k0, k1) and one more (the reader's
    // last record ends in it).  All of them are decoded speculatively at once; stitching - which needs where the chunk in front ended and
    // the 32 KB in front of that - starts when the owner of the slice in front has told (set_state; slice 0 knows), and tells the owner of
    // the next slice as soon as it has passed chunk k1 - 1 (end_state) - a chain of 32 KB hand-overs along the ranks while the expensive
    // part, the decoding, runs on all of them side by side.  The member's CRC is checked at the end, when the CRC of its bytes in front of
    // the slice is known (finish_crc).
    struct SliceState { uint64_t end_bit = 0; bool member_start = false, stop = false, bad = false, ready = false; std::vector<uint8_t> window; };
    struct SegRec { uint32_t crc = 0; uint64_t len = 0; bool has_end = false; uint32_t end_crc = 0, end_isize = 0; };
    size_t nchunks() const { return chunks_.size(); }
    bool setup()                                                     // header and chunk table (start() without the workers); false: not a gzip file this reader handles
    {
        const long h = member_header(base_, end_);
        if (h <= 0) return false;
        first_bit_ = (uint64_t)h * 8;
        const uint64_t data_bits = (uint64_t)(end_ - base_) * 8;
        for (uint64_t b = first_bit_; b < data_bits; b += (uint64_t)chunk_bytes_ * 8) { std::unique_ptr<Chunk> c(new Chunk()); c->nominal_bit = b; chunks_.push_back(std::move(c)); }
        lim_ = chunks_.size();
        return true;
    }
    bool start_slice(size_t k0, size_t k1)
    {
        if (!setup() || k0 >= k1 || k1 > chunks_.size()) return false;
        slice_ = true; slice_end_ = k1; lim_ = std::min(chunks_.size(), k1 + 1);
        shared_ = SharedBufs::get();
        consume_next_ = stitch_next_ = k0; next_decode_ = k0 == 0 ? 1 : k0; limit_decode_ = lim_;
        state_ready_ = k0 == 0;
        for (int i = 0; i < nthreads_; i++) workers_.emplace_back([this] { work(); });
        return true;
    }
    void set_state(const SliceState &st)                             // where the slice in front ended
    {
        stitched_end_bit_ = st.end_bit; at_member_start_ = st.member_start; window_ = st.window;
        if (st.stop) { stitch_stop_ = true; finished_ = true; }      // (the data ended in front of this slice: nothing to deliver)
        state_ready_ = true;
    }
    // stitches through the slice's last own chunk (waits for their decoding) and says where it ended; false: damaged data
    bool end_state(SliceState &out)
    {
        if (!state_ready_) return false;
        while (!end_state_.ready && !stitch_stop_ && stitch_next_ < slice_end_) stitch(slice_end_ - 1);
        if (!end_state_.ready) {                                     // (the data had ended in front of the slice)
            end_state_.end_bit = stitched_end_bit_; end_state_.member_start = at_member_start_; end_state_.window = window_; end_state_.stop = true; end_state_.ready = true;
        }
        out = end_state_;
        return !end_state_.bad;
    }
    // the slice's own text is [0, own_bytes()) of what read() delivers (the rest: the chunk behind it)
    // the member CRCs of the slice, given CRC and length of the open member's bytes in front of it; false: a member's CRC or length does not match
    static bool finish_crc(const std::vector<SegRec> &segs, size_t nsegs_own, uint32_t crc_in, uint64_t len_in, uint32_t *crc_out, uint64_t *len_out)
    {
        uint32_t rc = crc_in; uint64_t rl = len_in;
        for (size_t i = 0; i < nsegs_own && i < segs.size(); i++) {
            const SegRec &r = segs[i];
            if (r.len) { rc = rl ? (uint32_t)crc32_combine(rc, r.crc, (z_off_t)r.len) : r.crc; rl += r.len; }
            if (r.has_end) { if (rc != r.end_crc || (uint32_t)rl != r.end_isize) return false; rc = 0; rl = 0; }
        }
        *crc_out = rc; *len_out = rl;
        return true;
    }
    size_t segs() const { return segs_.size(); }
    const std::vector<SegRec> &seg_list() const { return segs_; }
    size_t next_chunk_index() const { return consume_next_; }
    bool slice_failed() const { return failed_; }
    const std::string &slice_error() const { return err_; }
    // appends the next chunk's bytes (nothing for a chunk the one in front ran over); false: none left, or the data is damaged (slice_failed)
    bool read_chunk(std::vector<uint8_t> &dst)
    {
        if (cur_) { release(cur_index_); cur_ = nullptr; }
        if (finished_) return false;
        if (!next_chunk()) return false;
        if (cur_ && cur_failed_) return false;
        if (cur_) dst.insert(dst.end(), cur_->bytes.data(), cur_->bytes.data() + cur_->bytes.size());
        return true;
    }

    // the next chunk's bytes where they lie (valid until the next call; n = 0 for a chunk the one in front ran over); false as read_chunk
    bool next_chunk_view(const uint8_t **p, size_t *n)
    {
        if (cur_) { release(cur_index_); cur_ = nullptr; }
        *p = nullptr; *n = 0;
        if (finished_) return false;
        if (!next_chunk()) return false;
        if (cur_ && cur_failed_) return false;
        if (cur_) { *p = cur_->bytes.data(); *n = cur_->bytes.size(); }
        return true;
    }

    // like a file read: < n only at the end of the data; bad: the file is truncated / damaged behind what was delivered
    int read(uint8_t *dst, int n, bool *bad, std::string *msg)
    {
        int got = 0;
        *bad = false;
        while (got < n) {
            if (cur_ && cur_off_ < cur_->bytes.size()) {
                const size_t k = std::min<size_t>((size_t)(n - got), cur_->bytes.size() - cur_off_);
                const uint64_t t0 = dbg_ ? now_ns() : 0;
                memcpy(dst + got, cur_->bytes.data() + cur_off_, k);
                if (dbg_) t_copy_ += now_ns() - t0;
                cur_off_ += k; got += (int)k;
                continue;
            }
            if (cur_) { if (cur_failed_) { *bad = true; *msg = err_; return got; } release(cur_index_); cur_ = nullptr; }
            if (finished_) { if (failed_) { *bad = true; *msg = err_; } return got; }
            if (!next_chunk()) { if (failed_ && !cur_) { *bad = true; *msg = err_; return got; } }
        }
        return got;
    }

private:
    const uint8_t *base_, *end_;
    int nthreads_;
    size_t chunk_bytes_;
    uint64_t first_bit_ = 0;
    std::vector<std::unique_ptr<Chunk>> chunks_;
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_work_, cv_done_;
    bool quit_ = false;
    size_t next_decode_ = 1;                                        // next chunk a worker decodes speculatively (chunk 0 is decoded from its known start)
    size_t limit_decode_ = 0;                                       // ... but none at or behind this one (bounds the memory in flight)
    std::deque<std::function<void()>> jobs_;                        // resolve jobs (they go first)
    // consumer state
    size_t cur_index_ = 0; Chunk *cur_ = nullptr; size_t cur_off_ = 0; bool cur_failed_ = false;
    size_t stitch_next_ = 0, consume_next_ = 0;                     // next chunk to stitch / to hand to the consumer
    size_t n_spec_ = 0, n_seq_ = 0, n_notfound_ = 0, n_skip_ = 0;
    std::atomic<uint64_t> t_decode_{0}, t_resolve_{0}, t_crc_{0};   // nanoseconds the workers spent (MC_PGZ_DEBUG prints them)
    uint64_t t_seq_ = 0, b_plain_ = 0, b_spec_ = 0, t_copy_ = 0, t_wait_ = 0;
    const bool dbg_ = getenv("MC_PGZ_DEBUG") != nullptr;
    static uint64_t now_ns() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec; }
    // Buffers of finished chunks are kept and handed to the next ones: a fresh 10 MB buffer is 2,500 page faults, and a dozen threads
    // faulting at once queue up in the kernel (the first 65 MB of a file took 0.85 s instead of 0.13 s).
    std::vector<std::unique_ptr<SymBuf>> pool_sym_;
    std::vector<ByteBuf> pool_bytes_;
    bool stitch_stop_ = false;                                      // the data ends (or is damaged) in the last stitched chunk
    uint64_t stitched_end_bit_ = 0;
    bool at_member_start_ = true;                                   // the stitched position is the first deflate bit of a member
    std::vector<uint8_t> window_;                                   // last <= 32 KB of the current member in front of the stitched position
    uint32_t run_crc_ = 0; uint64_t run_len_ = 0;                   // CRC / length of the current member so far
    bool finished_ = false, failed_ = false; std::string err_;
    size_t lim_ = 0;                                                // chunks [.., lim_) are decoded (all of them; a slice: its own and one more)
    bool slice_ = false, state_ready_ = true; size_t slice_end_ = 0; SliceState end_state_; std::vector<SegRec> segs_;
    std::shared_ptr<SharedBufs> shared_;

    void stop()
    {
        { std::unique_lock<std::mutex> lk(mu_); quit_ = true; cv_work_.notify_all(); }
        for (auto &t : workers_) t.join();
        workers_.clear();
    }

    void work()
    {
        Spec sp; sp.base = base_; sp.end = end_; sp.data_end_bit = (uint64_t)(end_ - base_) * 8;
        for (;;) {
            std::function<void()> job;
            size_t k = 0;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_work_.wait(lk, [&] { return quit_ || !jobs_.empty() || (next_decode_ < lim_ && next_decode_ < limit_decode_); });
                if (quit_) return;
                if (!jobs_.empty()) { job = std::move(jobs_.front()); jobs_.pop_front(); }
                else k = next_decode_++;
            }
            if (job) { job(); continue; }
            Chunk &c = *chunks_[k];
            take_buffers(c, true);
            const uint64_t stop_bit = k + 1 < chunks_.size() ? chunks_[k + 1]->nominal_bit : sp.data_end_bit;
            const uint64_t t0 = now_ns();
            try { c.found = sp.find_and_decode(c.nominal_bit, stop_bit, stop_bit, c); }
            catch (const std::bad_alloc &) { c.found = false; c.bytes.drop(); }   // (no memory for the speculative output: the chunk is decoded in turn, from its known start)
            t_decode_ += now_ns() - t0;
            if (!c.found) recycle_sym(c);
            { std::unique_lock<std::mutex> lk(mu_); c.state = 1; cv_done_.notify_all(); }
        }
    }

    void release(size_t k)
    {
        Chunk &c = *chunks_[k];
        recycle_sym(c);
        if (c.bytes.capacity()) {
            c.bytes.clear();
            std::unique_lock<std::mutex> lk(mu_);
            if (pool_bytes_.size() < 4 * (size_t)nthreads_) { pool_bytes_.emplace_back(); pool_bytes_.back().swap(c.bytes); }
        }
        if (shared_) shared_->give(c.bytes);
        c.bytes.drop();
    }
    void recycle_sym(Chunk &c)
    {
        if (!c.sym.cap) return;
        std::unique_lock<std::mutex> lk(mu_);
        if (pool_sym_.size() < 4 * (size_t)nthreads_) { pool_sym_.emplace_back(new SymBuf()); pool_sym_.back()->swap(c.sym); c.sym.n = 0; return; }
        lk.unlock();
        if (shared_) shared_->give(c.sym);
        c.sym.drop();
    }
    void take_buffers(Chunk &c, bool sym)
    {
        std::unique_lock<std::mutex> lk(mu_);
        if (sym && !c.sym.cap && !pool_sym_.empty()) { c.sym.swap(*pool_sym_.back()); pool_sym_.pop_back(); c.sym.n = 0; }
        if (!c.bytes.capacity() && !pool_bytes_.empty()) { c.bytes.swap(pool_bytes_.back()); pool_bytes_.pop_back(); c.bytes.clear(); }
        lk.unlock();
        if (shared_) {
            if (sym && !c.sym.cap) shared_->take(c.sym);
            if (!c.bytes.capacity()) shared_->take(c.bytes);
        }
    }

    void fail(const std::string &m) { failed_ = true; finished_ = true; err_ = m; }

    // accounts the bytes of a finished chunk to the running member CRC; false on a CRC / length mismatch
    // (a slice - start_slice - does not know the CRC of the member's bytes in front of it yet: it keeps its segments and checks them
    // when that arrives, finish_crc)
    bool account(Chunk &c)
    {
        if (slice_) {
            size_t at0 = 0;
            for (size_t s = 0; s <= c.ends.size(); s++) {
                const size_t to = s < c.ends.size() ? (size_t)c.ends[s].out_pos : c.bytes.size();
                const size_t n = to - at0;
                SegRec r; r.crc = n ? (s < c.seg_crc.size() ? c.seg_crc[s] : crc32_any(0, c.bytes.data() + at0, n)) : 0; r.len = n; r.has_end = s < c.ends.size();
                if (r.has_end) { r.end_crc = c.ends[s].crc; r.end_isize = c.ends[s].isize; }
                if (n || r.has_end) segs_.push_back(r);
                at0 = to;
            }
            return true;
        }
        size_t at = 0;
        for (size_t s = 0; s <= c.ends.size(); s++) {
            const size_t to = s < c.ends.size() ? (size_t)c.ends[s].out_pos : c.bytes.size();
            const size_t n = to - at;
            if (n) {
                const uint32_t sc = s < c.seg_crc.size() ? c.seg_crc[s] : crc32_any(0, c.bytes.data() + at, n);
                run_crc_ = run_len_ ? (uint32_t)crc32_combine(run_crc_, sc, (z_off_t)n) : sc;
                run_len_ += n;
            }
            if (s < c.ends.size()) {
                if (run_crc_ != c.ends[s].crc || (uint32_t)run_len_ != c.ends[s].isize) return false;
                run_crc_ = 0; run_len_ = 0;
            }
            at = to;
        }
        return true;
    }

    static void seg_crcs(Chunk &c)
    {
        c.seg_crc.clear();
        size_t at = 0;
        for (size_t s = 0; s <= c.ends.size(); s++) {
            const size_t to = s < c.ends.size() ? (size_t)c.ends[s].out_pos : c.bytes.size();
            uint32_t v = 0;
            v = crc32_any(0, c.bytes.data() + at, to - at);
            c.seg_crc.push_back(v);
            at = to;
        }
    }

    void update_window(const Chunk &c)
    {   // the last 32 KB of the current member behind this chunk
        size_t from = 0;
        bool fresh = false;
        if (!c.ends.empty()) { from = (size_t)c.ends.back().out_pos; fresh = true; }
        const size_t n = c.bytes.size() - from;
        if (fresh) window_.clear();
        if (n >= 32768) window_.assign(c.bytes.end() - 32768, c.bytes.end());
        else {
            window_.insert(window_.end(), c.bytes.begin() + (long)from, c.bytes.end());
            if (window_.size() > 32768) window_.erase(window_.begin(), window_.end() - 32768);
        }
    }

    // markers -> bytes for symbols [from, to) of a chunk whose window is w (wn bytes); false: a marker reaches in front of the member.
    // Through a table of all 256 + 32768 symbol values (byte values as themselves, marker k as byte k of the window, 0xFFFF where the
    // member has no such byte): one load per symbol and no branch.  The repetitive lines of a FASTQ file (qualities, '+', parts of the
    // names) stay markers for the whole chunk - each is copied from the record before it - so half of all symbols took the
    // branch of the per-symbol loop rounds 3 - 4 had behind their 16-at-a-time fast path.
    static bool resolve(const uint16_t *s, uint8_t *o, size_t from, size_t to, const uint8_t *w, size_t wn)
    {
        if (to - from < 4096) {                                      // (the 32 KB tail in front of a short stretch: not worth a table)
            for (size_t i = from; i < to; i++) {
                const uint16_t v = s[i];
                if (v < 256) o[i] = (uint8_t)v;
                else { const size_t j = (size_t)v - 256; if (j + wn < 32768) return false; o[i] = w[j + wn - 32768]; }
            }
            return true;
        }
        std::unique_ptr<uint16_t[]> lut(new uint16_t[256 + 32768]);
        for (int v = 0; v < 256; v++) lut[v] = (uint16_t)v;
        const size_t miss = 32768 - wn;                              // markers below this index point in front of the member
        for (size_t j = 0; j < miss; j++) lut[256 + j] = 0xFFFF;
        for (size_t j = miss; j < 32768; j++) lut[256 + j] = w[j - miss];
        uint32_t bad = 0;
        size_t i = from;
        for (; i + 8 <= to; i += 8) {
            const uint32_t a = lut[s[i]], b = lut[s[i + 1]], c = lut[s[i + 2]], d = lut[s[i + 3]], e = lut[s[i + 4]], f = lut[s[i + 5]], g = lut[s[i + 6]], h = lut[s[i + 7]];
            bad |= a | b | c | d | e | f | g | h;
            const uint64_t out8 = (uint64_t)(a & 255u) | ((uint64_t)(b & 255u) << 8) | ((uint64_t)(c & 255u) << 16) | ((uint64_t)(d & 255u) << 24) | ((uint64_t)(e & 255u) << 32) |
                                  ((uint64_t)(f & 255u) << 40) | ((uint64_t)(g & 255u) << 48) | ((uint64_t)(h & 255u) << 56);
            memcpy(o + i, &out8, 8);
        }
        for (; i < to; i++) { const uint32_t a = lut[s[i]]; bad |= a; o[i] = (uint8_t)a; }
        return (bad >> 8) == 0;
    }

    // Stitches the decoded chunks in file order as far as they are ready (at most `ahead` chunks in front of the consumer): checks
    // that a chunk starts where the one before it ended, hands its marker replacement + CRC to the workers, and keeps the 32 KB
    // window going; a chunk that does not fit is decoded here, sequentially.  wait_for: the chunk the consumer needs now.
    void stitch(size_t wait_for)
    {
        const uint64_t data_bits = (uint64_t)(end_ - base_) * 8;
        while (stitch_next_ < lim_ && stitch_next_ <= wait_for + (size_t)nthreads_ && !stitch_stop_) {
            const size_t k = stitch_next_;
            Chunk &c = *chunks_[k];
            const uint64_t from = k == 0 ? first_bit_ : stitched_end_bit_;
            const uint64_t stop_bit = k + 1 < chunks_.size() ? chunks_[k + 1]->nominal_bit : data_bits;
            if (k > 0) {
                std::unique_lock<std::mutex> lk(mu_);
                if (c.state < 1) { if (k > wait_for) return; cv_done_.wait(lk, [&] { return c.state >= 1; }); }
            }
            if (from >= stop_bit && k + 1 < chunks_.size()) {           // the chunk in front ran over this one entirely
                release(k); n_skip_++;
                c.bytes.clear(); c.ends.clear(); c.bad = false; c.at_eof = false; c.skipped = true;
                { std::unique_lock<std::mutex> lk(mu_); c.state = 2; }
                stitch_next_++;
                continue;
            }
            bool usable = k > 0 && c.found && c.start_bit == from && !(c.starts_member && !at_member_start_);
            if (getenv("MC_PGZ_DEBUG2")) fprintf(stderr, "chunk %zu nominal %llu from %llu found %d start %llu member %d at_member_start %d\n", k, (unsigned long long)c.nominal_bit, (unsigned long long)from, (int)c.found, (unsigned long long)c.start_bit, (int)c.starts_member, (int)at_member_start_);
            const size_t stat_total = c.total, stat_ns = c.sym.size();      // (read here: once the job below is queued, a worker owns the symbols)
            if (usable) {
                // the window behind this chunk needs its last 32 KB only: replaced here; the rest (and the CRC) is a job
                const size_t n = c.total, ns = c.sym.size();                 // the first ns symbols may hold markers, the rest are bytes already
                std::shared_ptr<std::vector<uint8_t>> w(new std::vector<uint8_t>(c.starts_member ? std::vector<uint8_t>() : window_));
                c.bytes.resize(n);
                const size_t tail = n > 32768 ? n - 32768 : 0;
                size_t mfrom = c.ends.empty() ? 0 : (size_t)c.ends.back().out_pos;      // the current member's part of the chunk
                if (std::max(tail, mfrom) < ns && !resolve(c.sym.data(), c.bytes.data(), std::max(tail, mfrom), ns, w->data(), w->size())) usable = false;
                else {
                    std::vector<uint8_t> nw;
                    if (!c.ends.empty()) window_.clear();
                    if (n - mfrom >= 32768) nw.assign(c.bytes.begin() + (long)(n - 32768), c.bytes.end());   // (the tail has just been resolved above)
                    else { nw = window_; nw.insert(nw.end(), c.bytes.begin() + (long)mfrom, c.bytes.end()); if (nw.size() > 32768) nw.erase(nw.begin(), nw.end() - 32768); }
                    window_.swap(nw);
                    Chunk *cp = &c;
                    std::unique_lock<std::mutex> lk(mu_);
                    jobs_.push_back([this, cp, w] {
                        const uint64_t t0 = now_ns();
                        const bool ok = resolve(cp->sym.data(), cp->bytes.data(), 0, cp->sym.size(), w->data(), w->size());
                        recycle_sym(*cp);
                        if (!ok) { cp->bad = true; cp->msg = "invalid distance too far back"; cp->bytes.clear(); cp->ends.clear(); }
                        const uint64_t t1 = now_ns();
                        try { seg_crcs(*cp); } catch (const std::bad_alloc &) { cp->bad = true; cp->msg = "out of memory"; cp->bytes.clear(); cp->ends.clear(); cp->seg_crc.clear(); }
                        t_resolve_ += t1 - t0; t_crc_ += now_ns() - t1;
                        std::unique_lock<std::mutex> lk2(mu_);
                        cp->state = 2;
                        cv_done_.notify_all();
                    });
                    cv_work_.notify_all();
                }
            }
            if (usable) { n_spec_++; b_spec_ += stat_total; b_plain_ += stat_total - stat_ns; } else { n_seq_++; if (k > 0 && !c.found) n_notfound_++; }
            if (!usable) {                                               // the sequential path: from the known position with the known window
                recycle_sym(c);
                if (k == 0) take_buffers(c, false);
                c.ends.clear(); c.at_eof = false;
                const uint64_t t0 = now_ns();
                decode_known(base_, end_, from, stop_bit, at_member_start_ ? nullptr : window_.data(), at_member_start_ ? 0 : window_.size(), c);
                seg_crcs(c);
                t_seq_ += now_ns() - t0;
                update_window(c);
                std::unique_lock<std::mutex> lk(mu_);
                c.state = 2;
            }
            stitched_end_bit_ = c.end_bit;
            at_member_start_ = !c.ends.empty() && (size_t)c.ends.back().out_pos == (usable ? c.bytes.size() : c.bytes.size());
            stitch_next_ = k + 1;
            if (c.bad || c.at_eof || c.end_bit >= data_bits) stitch_stop_ = true;   // nothing behind this chunk
            if (slice_ && (k + 1 == slice_end_ || stitch_stop_) && !end_state_.ready) {   // (a slice: what the owner of the next one needs - start_slice)
                end_state_.end_bit = stitched_end_bit_; end_state_.member_start = at_member_start_;
                end_state_.window = window_; end_state_.stop = stitch_stop_; end_state_.bad = c.bad; end_state_.ready = true;
            }
        }
    }

    // makes the next chunk's bytes available as cur_; false when the stream has ended
    bool next_chunk()
    {
        const size_t k = consume_next_;
        if (k >= lim_ || (stitch_stop_ && k >= stitch_next_)) { finished_ = true; return false; }
        {   // let the workers run ahead of the consumer (bounded)
            std::unique_lock<std::mutex> lk(mu_);
            limit_decode_ = std::max(limit_decode_, k + 1 + (size_t)nthreads_ * 2);
            cv_work_.notify_all();
        }
        stitch(k);
        Chunk &c = *chunks_[k];
        { const uint64_t t0 = dbg_ ? now_ns() : 0; std::unique_lock<std::mutex> lk(mu_); cv_done_.wait(lk, [&] { return c.state >= 2; }); if (dbg_) t_wait_ += now_ns() - t0; }
        consume_next_ = k + 1;
        if (c.skipped) return true;
        const uint64_t data_bits = (uint64_t)(end_ - base_) * 8;
        if (!account(c)) { c.bytes.clear(); fail("CRC check failed"); return false; }
        cur_ = &c; cur_index_ = k; cur_off_ = 0; cur_failed_ = false;
        if (c.bad) { cur_failed_ = true; failed_ = true; finished_ = true; err_ = c.msg; }
        else if (c.at_eof) finished_ = true;
        else if (c.end_bit >= data_bits) { cur_failed_ = true; failed_ = true; finished_ = true; err_ = "compressed file ended before the end-of-stream marker was reached"; }
        return true;
    }
};

}   // namespace mcgz
