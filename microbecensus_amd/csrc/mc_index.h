// mc_index.h - host-side construction of the marker index and of the constant tables.
//
// Replaces prerapsearch (CHashSearch::BuildDHash@0x40fc20 in
// /root/reference/microbe_census/bin/prerapsearch_Linux_2.15, same code as in rapsearch_Linux_2.15) and
// the DB/statistics set-up of CHashSearch::Search@0x418750.  The index has to reproduce the posting ORDER
// inside every bucket, because the order in which seed hits are met decides which of several equivalent
// HSPs survives; prerapsearch orders a bucket with std::sort(CompDbObj) - an unstable sort - so the
// libstdc++ (GCC 4.4) introsort is restated here.  tests/test_index.py checks the result entry by entry
// against a prerapsearch-built database.
#pragma once
#include <sys/stat.h>
#include <unistd.h>
#include "mc_core.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <algorithm>

struct McHostIndex {
    std::vector<std::string> names;
    std::vector<uint8_t> res_code;     // RAPsearch2 codes (group<<4 | 1+k, 0xA0 invalid) - for cross-checks
    std::vector<uint8_t> res;          // dense codes used on the device
    std::vector<uint32_t> off, bstart, post;
    std::vector<uint16_t> keys;
    std::vector<uint32_t> bitmap;      // 1 bit per bucket: non-empty
    std::vector<McBucketRec> rec;      // first-residue group boundaries per bucket; empty when the index cannot use them
    std::vector<uint32_t> filt;        // Bloom filter over (bucket, 3-residue key): the exact 9-mer probes
    std::vector<uint32_t> wild;        // wildcard filter over the 10-mers (mc_wild_*)
    std::vector<uint32_t> pair;        // pair filter over the 10-mers (mc_pair_*)
    std::vector<unsigned long long> rt; // range table of the long first-residue groups (mc_rt_*)
    uint32_t rt_mask;
    uint32_t max_bucket;
    uint32_t freq_thr;
    double letter_p[10];
    int64_t nres;
    int nseq;
};

static const char *MC_AA_ORDER = "ARNDCQEGHILKMFPSTWYV";
static const char *MC_GROUPS[10] = {"A", "KR", "EDNQ", "C", "G", "H", "ILVM", "FYW", "P", "S/T"};

inline int mc_dense_of_char(unsigned char c)
{
    if (c >= 'a' && c <= 'z') c = (unsigned char)(c - 32);
    const char *p = c ? strchr(MC_AA_ORDER, c) : NULL;
    return p ? (int)(p - MC_AA_ORDER) : MC_INV;
}
inline int mc_group_of_dense(int d)
{
    if (d >= 20) return MC_INVGRP;
    for (int g = 0; g < 10; g++) if (strchr(MC_GROUPS[g], MC_AA_ORDER[d])) return g;
    return MC_INVGRP;
}
inline int mc_code_of_char(unsigned char c)
{ // CHashSearch ctor 0x4169bd-0x416a62
    if (c >= 'a' && c <= 'z') c = (unsigned char)(c - 32);
    for (int g = 0; g < 10; g++) { const char *p = c ? strchr(MC_GROUPS[g], c) : NULL; if (p) return (g << 4) + 1 + (int)(p - MC_GROUPS[g]); }
    return 0xA0;
}

// ---- libstdc++ std::sort as prerapsearch's build instantiates it (threshold 16, heap fallback) ------------
template <class V, class Less> void mc44_adjust_heap(V *first, long hole, long len, V value, Less lt)
{
    long top = hole, sc = hole;
    while (sc < (len - 1) / 2) { sc = 2 * (sc + 1); if (lt(first[sc], first[sc - 1])) sc--; first[hole] = first[sc]; hole = sc; }
    if ((len & 1) == 0 && sc == (len - 2) / 2) { sc = 2 * (sc + 1); first[hole] = first[sc - 1]; hole = sc - 1; }
    long parent = (hole - 1) / 2;
    while (hole > top && lt(first[parent], value)) { first[hole] = first[parent]; hole = parent; parent = (hole - 1) / 2; }
    first[hole] = value;
}
template <class V, class Less> void mc44_introsort_loop(V *first, V *last, long depth, Less lt)
{
    while (last - first > 16) {
        if (depth == 0) {
            long n = last - first;
            for (long parent = (n - 2) / 2;; parent--) { mc44_adjust_heap(first, parent, n, first[parent], lt); if (parent == 0) break; }
            for (long m = n; m > 1;) { m--; V v = first[m]; first[m] = first[0]; mc44_adjust_heap(first, 0L, m, v, lt); }
            return;
        }
        --depth;
        // prerapsearch_Linux_2.15 was built with a libstdc++ that moves the median of (first, mid, last-1) to *first
        // and partitions [first+1, last) around it (__move_median_first, GCC 4.5-4.8); verified bucket by bucket
        // against the database prerapsearch wrote (tests/test_emul.py::test_index_builder_matches_prerapsearch).
        V *x = first, *y = first + (last - first) / 2, *z = last - 1;
        if (lt(*x, *y)) { if (lt(*y, *z)) std::swap(*x, *y); else if (lt(*x, *z)) std::swap(*x, *z); }
        else if (lt(*x, *z)) {}
        else if (lt(*y, *z)) std::swap(*x, *z);
        else std::swap(*x, *y);
        V pivot = *first;
        V *lo = first + 1, *hi = last;
        for (;;) {
            while (lt(*lo, pivot)) ++lo;
            --hi;
            while (lt(pivot, *hi)) --hi;
            if (!(lo < hi)) break;
            V t = *lo; *lo = *hi; *hi = t;
            ++lo;
        }
        mc44_introsort_loop(lo, last, depth, lt);
        last = lo;
    }
}
template <class V, class Less> void mc44_sort(V *first, V *last, Less lt)
{
    long n = last - first, lg = 0;
    if (n <= 0) return;
    for (long t = n; t > 1; t >>= 1) lg++;
    mc44_introsort_loop(first, last, 2 * lg, lt);
    V *stop = (n > 16) ? first + 16 : last;
    for (V *i = first + 1; i < stop; ++i) {
        V val = *i;
        if (lt(val, *first)) { for (V *p = i; p != first; --p) *p = *(p - 1); *first = val; }
        else { V *l = i, *nx = i - 1; while (lt(val, *nx)) { *l = *nx; l = nx; --nx; } *l = val; }
    }
    for (V *i = stop; i < last; ++i) { V val = *i, *l = i, *nx = i - 1; while (lt(val, *nx)) { *l = *nx; l = nx; --nx; } *l = val; }
}

struct McPostKey { uint32_t post; uint16_t key; int16_t rem; };   // rem = residues left after the 6-mer (uncapped)

inline void mc_index_derive(McHostIndex &X);

inline bool mc_build_index(McHostIndex &X, const char *const *names, const char *const *seqs, int nseq, std::string &err)
{
    X.nseq = nseq;
    X.names.assign(names, names + nseq);
    X.off.assign((size_t)nseq + 1, 0);
    int64_t total = 0;
    for (int s = 0; s < nseq; s++) {
        size_t l = strlen(seqs[s]);
        if (l >= 2048) { err = "marker sequence longer than 2047 residues: " + X.names[s]; return false; }
        X.off[s] = (uint32_t)total; total += (int64_t)l;
    }
    X.off[nseq] = (uint32_t)total; X.nres = total;
    X.res.resize((size_t)total); X.res_code.resize((size_t)total);
    uint8_t grp_of_dense[32];
    for (int d = 0; d < 32; d++) grp_of_dense[d] = (uint8_t)mc_group_of_dense(d);
    for (int s = 0; s < nseq; s++)
        for (uint32_t i = X.off[s], k = 0; i < X.off[s + 1]; i++, k++) {
            unsigned char c = (unsigned char)seqs[s][k];
            X.res[i] = (uint8_t)mc_dense_of_char(c);
            X.res_code[i] = (uint8_t)mc_code_of_char(c);
        }
    // buckets: every window but the last of each sequence, 6 valid reduced residues
    std::vector<uint32_t> count(MC_NBUCKET + 1, 0);
    for (int pass = 0; pass < 2; pass++) {
        if (pass == 1) {
            X.bstart.assign(MC_NBUCKET + 1, 0);
            for (int b = 0; b < MC_NBUCKET; b++) X.bstart[b + 1] = X.bstart[b] + count[b];
            X.post.resize(X.bstart[MC_NBUCKET]);
            std::fill(count.begin(), count.end(), 0);
        }
        for (int s = 0; s < nseq; s++) {
            const uint8_t *r = &X.res[X.off[s]];
            int len = (int)(X.off[s + 1] - X.off[s]);
            for (int pos = 0; pos + 6 < len; pos++) {
                int seed = 0; bool bad = false;
                for (int k = 0; k < 6; k++) { int g = grp_of_dense[r[pos + k]]; if (g == MC_INVGRP) { bad = true; break; } seed = seed * 10 + g; }
                if (bad) continue;
                if (pass == 1) X.post[X.bstart[seed] + count[seed]] = ((uint32_t)s << 11) | (uint32_t)pos;
                count[seed]++;
            }
        }
    }
    // suffix keys + bucket order (std::sort with CompDbObj, __introsort_loop@0x42f8d0)
    X.keys.assign(X.post.size() + 64, 0xFFFF);   // +64: mc_group_range8 reads whole words around a group
    X.bitmap.assign((MC_NBUCKET + 31) / 32, 0);
    std::vector<McPostKey> tmp;
    auto key_of = [&](uint32_t p) -> uint16_t {
        int s = (int)(p >> 11), pos = (int)(p & 0x7ff), len = (int)(X.off[s + 1] - X.off[s]);
        int rem = len - pos - 6; if (rem > 4) rem = 4;
        uint32_t k = 0;
        for (int i = 0; i < 4; i++) k |= (uint32_t)(i < rem ? grp_of_dense[X.res[X.off[s] + pos + 6 + i]] : 0xF) << (12 - 4 * i);
        return (uint16_t)k;
    };
    // CompDbObj (inlined in __introsort_loop@0x42f8d0, e.g. 0x42fa5a-0x42fae1): compare the reduced residues that
    // follow the 6-mer over n = 4 (both suffixes longer than 3) or min(remA, remB) positions; on a tie the
    // posting with FEWER residues left in its sequence sorts first.
    auto less = [](const McPostKey &a, const McPostKey &b) {
        int n = (a.rem > 3 && b.rem > 3) ? 4 : (a.rem < b.rem ? a.rem : b.rem);
        if (n > 0) { int sh = (4 - n) * 4; int x = a.key >> sh, y = b.key >> sh; if (x != y) return x < y; }
        return a.rem < b.rem;
    };
    for (int b = 0; b < MC_NBUCKET; b++) {
        uint32_t n = X.bstart[b + 1] - X.bstart[b];
        if (!n) continue;
        X.bitmap[b >> 5] |= 1u << (b & 31);
        tmp.resize(n);
        for (uint32_t i = 0; i < n; i++) { uint32_t p = X.post[X.bstart[b] + i]; int sq = (int)(p >> 11); tmp[i].post = p; tmp[i].key = key_of(p); tmp[i].rem = (int16_t)((int)(X.off[sq + 1] - X.off[sq]) - (int)(p & 0x7ff) - 6); }
        mc44_sort(tmp.data(), tmp.data() + n, less);
        for (uint32_t i = 0; i < n; i++) { X.post[X.bstart[b] + i] = tmp[i].post; X.keys[X.bstart[b] + i] = tmp[i].key; }
    }
    mc_index_derive(X);
    return true;
}

// Everything that follows from (res, off, bstart, post, keys): bucket bitmap, bucket records, filters, range table and the
// two figures of the .info file.  Shared by the FASTA builder above and the rapdb loader below.
inline void mc_index_derive(McHostIndex &X)
{
    X.bitmap.assign((MC_NBUCKET + 31) / 32, 0);
    for (int b = 0; b < MC_NBUCKET; b++) if (X.bstart[b + 1] > X.bstart[b]) X.bitmap[b >> 5] |= 1u << (b & 31);
    // first-residue group boundaries (McBucketRec).  Valid only if every bucket really is grouped that way and fits 16 bit.
    {
        X.rec.assign(MC_NBUCKET, McBucketRec());
        X.max_bucket = 0;
        for (int b = 0; b < MC_NBUCKET; b++) X.max_bucket = std::max(X.max_bucket, X.bstart[b + 1] - X.bstart[b]);
        bool ok = true;
        for (int b = 0; b < MC_NBUCKET && ok; b++) {
            const uint32_t b0 = X.bstart[b], n = X.bstart[b + 1] - b0;
            McBucketRec &R = X.rec[b];
            memset(&R, 0, sizeof R);
            R.start = b0;
            if (n > 0xFFFF) { ok = false; break; }
            int prev = -1;                                     // group of the previous posting: -1 = key FFFF, else first residue
            uint32_t cnt[13]; memset(cnt, 0, sizeof cnt);
            for (uint32_t i = 0; i < n; i++) {
                const uint32_t k = X.keys[b0 + i];
                const int g = (k == 0xFFFF) ? -1 : (int)(k >> 12);
                if (g < prev || g > 10) { if (getenv("MC_DEBUG_REC")) fprintf(stderr, "rec: bucket %d posting %u key %04x after group %d (n %u)\n", b, i, k, prev, n); ok = false; break; }
                prev = g;
                cnt[g + 1]++;                                  // cnt[0] = sequence-end postings, cnt[1 + g] = group g
            }
            uint32_t run = cnt[0];
            for (int g = 0; g < 11; g++) { R.cum[g] = (uint16_t)run; run += cnt[1 + g]; }   // group 10 = the invalid residue
            R.cum[11] = (uint16_t)n;
        }
        if (!ok) X.rec.clear();
    }
    // 9-mer filter, and the wildcard and pair filters over the postings whose key has 4 residues
    X.filt.assign(MC_FILT9_WORDS, 0);
    X.wild.assign((size_t)MC_WILD_LINES * MC_WILD_LINE_WORDS, 0);
    X.pair.assign((size_t)MC_PAIR_BLOCKS * 4, 0);
    for (int b = 0; b < MC_NBUCKET; b++)
        for (uint32_t i = X.bstart[b]; i < X.bstart[b + 1]; i++) {
            const uint32_t k = X.keys[i];
            if ((k & 0xF0) != 0xF0) {                          // at least 3 key residues: inside the range of its 9-mer probe
                const uint32_t h9 = mc_filter_hash((uint32_t)b, k | 0xFu);
                X.filt[mc_filter9_word(h9)] |= mc_filter_bits(h9);
            }
            if ((k & 0xF) == 0xF) continue;                    // shorter key: never inside the range of a 10-mer probe
            const uint32_t ctx = mc_wild_ctx((uint32_t)b, k), line = mc_wild_line(ctx);
            for (int g = 0; g < 4; g++) {
                mc_wild_set(&X.wild[(size_t)line * MC_WILD_LINE_WORDS + (size_t)g * 2], mc_wild_bits(ctx, (uint32_t)b, k, g));
                const uint32_t hp = mc_pair_hash((uint32_t)b, k, g);
                mc_pair_set(&X.pair[(size_t)mc_pair_block(hp) * 4], hp, mc_pair_digit((uint32_t)b, k, g));
            }
        }
    // range table: every query key with a range inside a first-residue group of more than 8 postings, with the range
    // mc_key_range returns for it
    {
        McIndex V; V.res = X.res.data(); V.off = X.off.data(); V.bstart = X.bstart.data(); V.post = X.post.data(); V.keys = X.keys.data(); V.rec = nullptr; V.nseq = X.nseq;
        std::vector<unsigned long long> ent;
        std::vector<uint32_t> cand;
        for (int b = 0; b < MC_NBUCKET && !X.rec.empty(); b++) {
            const McBucketRec &R = X.rec[b];
            for (int g = 0; g < 11; g++) {
                const uint32_t g0 = R.cum[g], g1 = R.cum[g + 1];
                if (g1 - g0 <= 8) continue;
                cand.clear();
                for (uint32_t i = g0; i < g1; i++) {
                    const uint32_t k = X.keys[R.start + i];
                    if ((k & 0xF) != 0xF) cand.push_back(k);                  // 10-mer probe form
                    if ((k & 0xF0) != 0xF0) cand.push_back(k | 0xFu);          // 9-mer probe form
                }
                std::sort(cand.begin(), cand.end());
                cand.erase(std::unique(cand.begin(), cand.end()), cand.end());
                for (uint32_t qk : cand) {
                    McSeedCount sc{0, 0, 0};
                    int nst = 0;
                    const int cnt = mc_key_range(V, b, qk, &nst, &sc);
                    if (cnt > 0) ent.push_back(mc_rt_pack((uint32_t)b, qk, (uint32_t)nst, (uint32_t)cnt));
                }
            }
        }
        uint32_t cap = 1024;
        while (cap < 2 * ent.size() + 16) cap <<= 1;
        X.rt.assign(cap, ~0ull);
        X.rt_mask = cap - 1;
        for (unsigned long long e : ent) {
            uint32_t i = mc_rt_hash((uint32_t)(e >> 38), (uint32_t)((e >> 22) & 0xFFFF)) & X.rt_mask;
            while (X.rt[i] != ~0ull) i = (i + 1) & X.rt_mask;
            X.rt[i] = e;
        }
    }
    // .info: median of ALL bucket sizes, reduced-letter frequencies
    { std::vector<uint32_t> c(MC_NBUCKET); for (int b = 0; b < MC_NBUCKET; b++) c[b] = X.bstart[b + 1] - X.bstart[b]; std::nth_element(c.begin(), c.begin() + (MC_NBUCKET >> 1), c.end()); X.freq_thr = c[MC_NBUCKET >> 1]; }
    {
        int64_t gcount[11] = {0}, valid = 0;
        for (uint8_t d : X.res) gcount[mc_group_of_dense(d)]++;
        valid = (int64_t)X.res.size();                     // prerapsearch divides by ALL residues, the few invalid ones included
        for (int g = 0; g < 10; g++) X.letter_p[g] = (double)gcount[g] / (double)valid;
    }
}


// ---- loading a database written by `prerapsearch -d markers.faa -n rapdb` -------------------------------
// boost::archive::binary_oarchive, as RAPSearch2 2.15 writes it (little endian, 64-bit counts):
//   0x28 bytes of archive header | u64 n + n residue codes (group << 4 | 1 + index inside the group, 0xA0 = none) |
//   u64 nseq+1 + offsets (u32) | 5 + 12 bytes | 10^6 x (u64 count + postings u32: seq << 11 | pos) |
//   5 + 12 bytes | nseq x (u64 length + name) | 5 + 12 bytes | 10^6 x (u64 count + suffix keys u16)
// The postings and keys are taken in the file's order - which is the order mc_build_index reproduces from the FASTA.
inline bool mc_load_rapdb(McHostIndex &X, const char *path, std::string &err)
{
    FILE *f = fopen(path, "rb");
    if (!f) { err = std::string("cannot open ") + path; return false; }
    std::vector<uint8_t> b;
    { fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET); b.resize(n > 0 ? (size_t)n : 0); if (n > 0 && fread(b.data(), 1, (size_t)n, f) != (size_t)n) { fclose(f); err = "short read"; return false; } fclose(f); }
    size_t p = 0x28;
    bool bad = false;
    auto u64 = [&](void) -> uint64_t { uint64_t v = 0; if (p + 8 > b.size()) { bad = true; return 0; } memcpy(&v, &b[p], 8); p += 8; return v; };
    auto need = [&](uint64_t n) { if (p + n > b.size()) bad = true; return !bad; };
    if (b.size() < 0x40 || memcmp(&b[8], "serialization::archive", 22) != 0) { err = "not a boost binary archive"; return false; }
    const uint64_t nres = u64();
    if (bad || !need(nres)) { err = "truncated rapdb (residues)"; return false; }
    X.res_code.assign(b.begin() + (long)p, b.begin() + (long)(p + nres)); p += nres;
    const uint64_t noff = u64();
    if (bad || noff < 1 || !need(4 * noff)) { err = "truncated rapdb (offsets)"; return false; }
    X.nseq = (int)noff - 1; X.nres = (int64_t)nres;
    X.off.resize(noff); memcpy(X.off.data(), &b[p], 4 * noff); p += 4 * noff;
    p += 5;
    if (u64() != MC_NBUCKET || bad) { err = "rapdb: unexpected bucket count"; return false; }
    p += 4;
    X.bstart.assign(MC_NBUCKET + 1, 0);
    X.post.clear();
    for (int i = 0; i < MC_NBUCKET; i++) {
        const uint64_t c = u64();
        if (bad || !need(4 * c)) { err = "truncated rapdb (postings)"; return false; }
        const size_t o = X.post.size();
        X.post.resize(o + c);
        if (c) memcpy(&X.post[o], &b[p], 4 * c);
        p += 4 * c;
        X.bstart[i + 1] = (uint32_t)X.post.size();
    }
    p += 5;
    if (u64() != (uint64_t)X.nseq || bad) { err = "rapdb: name count mismatch"; return false; }
    p += 4;
    X.names.clear();
    for (int i = 0; i < X.nseq; i++) { const uint64_t l = u64(); if (bad || !need(l)) { err = "truncated rapdb (names)"; return false; } X.names.emplace_back((const char *)&b[p], (size_t)l); p += l; }
    p += 5 + 8 + 4;
    X.keys.assign(X.post.size() + 64, 0xFFFF);
    for (int i = 0; i < MC_NBUCKET; i++) {
        const uint64_t c = u64();
        if (bad || c != X.bstart[i + 1] - X.bstart[i] || !need(2 * c)) { err = "truncated rapdb (keys)"; return false; }
        if (c) memcpy(&X.keys[X.bstart[i]], &b[p], 2 * c);
        p += 2 * c;
    }
    // residue codes -> dense codes
    X.res.resize(X.res_code.size());
    for (size_t i = 0; i < X.res.size(); i++) {
        const int c = X.res_code[i], g = c >> 4, k = (c & 15) - 1;
        X.res[i] = (g < 10 && k >= 0 && k < (int)strlen(MC_GROUPS[g])) ? (uint8_t)mc_dense_of_char((unsigned char)MC_GROUPS[g][k]) : (uint8_t)MC_INV;
    }
    for (int s2 = 0; s2 < X.nseq; s2++) if (X.off[s2 + 1] - X.off[s2] >= 2048) { err = "marker sequence longer than 2047 residues: " + X.names[s2]; return false; }
    mc_index_derive(X);
    return true;
}

// ---- and writing one: the two files `prerapsearch -d markers.faa -n <path>` leaves (<path> and <path>.info) ------------
inline bool mc_write_rapdb(const McHostIndex &X, const char *path, std::string &err)
{
    static const unsigned char hdr[0x28] = {0x16, 0, 0, 0, 0, 0, 0, 0, 's', 'e', 'r', 'i', 'a', 'l', 'i', 'z', 'a', 't', 'i', 'o', 'n', ':', ':', 'a', 'r', 'c', 'h', 'i', 'v', 'e',
                                            0x09, 0x00, 0x04, 0x08, 0x04, 0x08, 0x01, 0x00, 0x00, 0x00};
    std::vector<uint8_t> b;
    auto put = [&](const void *p, size_t n) { const uint8_t *q = (const uint8_t *)p; b.insert(b.end(), q, q + n); };
    auto u64 = [&](uint64_t v) { put(&v, 8); };
    auto u32 = [&](uint32_t v) { put(&v, 4); };
    auto cls = [&](uint64_t count) { const uint8_t z[5] = {0, 0, 0, 0, 0}; put(z, 5); u64(count); u32(0); };   // class info + count + item version
    put(hdr, sizeof hdr);
    u64((uint64_t)X.res_code.size()); put(X.res_code.data(), X.res_code.size());
    u64((uint64_t)X.off.size()); put(X.off.data(), X.off.size() * 4);
    cls(MC_NBUCKET);
    for (int i = 0; i < MC_NBUCKET; i++) { const uint32_t c = X.bstart[i + 1] - X.bstart[i]; u64(c); if (c) put(&X.post[X.bstart[i]], (size_t)c * 4); }
    cls((uint64_t)X.nseq);
    for (int i = 0; i < X.nseq; i++) { u64((uint64_t)X.names[(size_t)i].size()); put(X.names[(size_t)i].data(), X.names[(size_t)i].size()); }
    cls(MC_NBUCKET);
    for (int i = 0; i < MC_NBUCKET; i++) { const uint32_t c = X.bstart[i + 1] - X.bstart[i]; u64(c); if (c) put(&X.keys[X.bstart[i]], (size_t)c * 2); }
    FILE *f = fopen(path, "wb");
    if (!f || fwrite(b.data(), 1, b.size(), f) != b.size()) { if (f) fclose(f); err = std::string("cannot write ") + path; return false; }
    fclose(f);
    // .info: header, 1, nseq, nres, bucket sizes, their median (the seed-lengthening threshold), the 10 group frequencies
    b.clear();
    put(hdr, sizeof hdr);
    u32(1); u64((uint64_t)X.nseq); u64((uint64_t)X.res_code.size()); u64(MC_NBUCKET);
    for (int i = 0; i < MC_NBUCKET; i++) u32(X.bstart[i + 1] - X.bstart[i]);
    u32(X.freq_thr); u64(10); put(X.letter_p, 80);
    const std::string ip = std::string(path) + ".info";
    f = fopen(ip.c_str(), "wb");
    if (!f || fwrite(b.data(), 1, b.size(), f) != b.size()) { if (f) fclose(f); err = "cannot write " + ip; return false; }
    fclose(f);
    return true;
}

// ---- the built index as one file (the per-user cache of mc_open) ------------------------------------------------------------
// mc_build_index takes half a second of host time per handle (buckets, suffix keys, filters, range table) - more than the search of
// the reference's default run of 1 - 2 M reads.  The result is a pure function of the marker names and sequences, so it is kept in a
// file named by a hash of those: header (magic, layout version, the hash, counts), the arrays as they lie in memory, a checksum of
// the payload.  A file that does not match in every respect is ignored and rebuilt; it is written to a temporary name and renamed.
#define MC_IXC_MAGIC 0x3158494D434D4D43ull      // "CMMCMIX1"
#define MC_IXC_VERSION 5u
inline uint64_t mc_ixc_input_hash(const char *const *names, const char *const *seqs, int nseq)
{
    uint64_t h = 0xCBF29CE484222325ull ^ (uint64_t)MC_IXC_VERSION;
    for (int s = 0; s < nseq; s++) {
        for (const char *p = names[s]; *p; p++) { h ^= (uint8_t)*p; h *= 0x100000001B3ull; }
        h ^= 0xFF; h *= 0x100000001B3ull;
        for (const char *p = seqs[s]; *p; p++) { h ^= (uint8_t)*p; h *= 0x100000001B3ull; }
        h ^= 0xFE; h *= 0x100000001B3ull;
    }
    return h;
}
inline uint64_t mc_ixc_sum(const void *p, size_t bytes, uint64_t acc)
{   // order-dependent sum of 64-bit words (the tail zero-padded): catches truncation and bit rot, not an adversary (the cache directory is the user's own)
    const uint8_t *b = (const uint8_t *)p;
    size_t i = 0;
    for (; i + 8 <= bytes; i += 8) { uint64_t w; memcpy(&w, b + i, 8); acc = (acc ^ w) * 0x9E3779B97F4A7C15ull + (acc >> 29); }
    if (i < bytes) { uint64_t w = 0; memcpy(&w, b + i, bytes - i); acc = (acc ^ w) * 0x9E3779B97F4A7C15ull + (acc >> 29); }
    return acc;
}
struct McIxcHeader { uint64_t magic; uint32_t version, nseq; uint64_t input_hash; uint64_t n[12]; uint32_t rt_mask, max_bucket, freq_thr, pad; double letter_p[10]; int64_t nres; uint64_t names_bytes; };
#define MC_IXC_ARRAYS(X, F)                                                                                                                   \
    F(0, (X).res_code) F(1, (X).res) F(2, (X).off) F(3, (X).bstart) F(4, (X).post) F(5, (X).keys) F(6, (X).bitmap) F(7, (X).rec) F(8, (X).filt) \
    F(9, (X).wild) F(10, (X).pair) F(11, (X).rt)
inline bool mc_index_save(const McHostIndex &X, uint64_t input_hash, const char *path)
{
    McIxcHeader H; memset(&H, 0, sizeof H);
    H.magic = MC_IXC_MAGIC; H.version = MC_IXC_VERSION; H.nseq = (uint32_t)X.nseq; H.input_hash = input_hash;
#define MC_F(i, v) H.n[i] = (uint64_t)(v).size();
    MC_IXC_ARRAYS(X, MC_F)
#undef MC_F
    H.rt_mask = X.rt_mask; H.max_bucket = X.max_bucket; H.freq_thr = X.freq_thr; H.nres = X.nres;
    for (int g = 0; g < 10; g++) H.letter_p[g] = X.letter_p[g];
    std::string names;
    for (const std::string &nm : X.names) { names += nm; names.push_back('\0'); }
    H.names_bytes = names.size();
    const std::string tmp = std::string(path) + ".tmp" + std::to_string((long)getpid());
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return false;
    uint64_t sum = mc_ixc_sum(&H, sizeof H, 0);
    bool ok = fwrite(&H, sizeof H, 1, f) == 1;
    ok = ok && (names.empty() || fwrite(names.data(), 1, names.size(), f) == names.size());
    sum = mc_ixc_sum(names.data(), names.size(), sum);
#define MC_F(i, v) { const size_t b_ = (v).size() * sizeof((v)[0]); ok = ok && (b_ == 0 || fwrite((v).data(), 1, b_, f) == b_); sum = mc_ixc_sum((v).data(), b_, sum); }
    MC_IXC_ARRAYS(X, MC_F)
#undef MC_F
    ok = ok && fwrite(&sum, 8, 1, f) == 1;
    ok = (fclose(f) == 0) && ok;
    if (!ok || rename(tmp.c_str(), path) != 0) { remove(tmp.c_str()); return false; }
    return true;
}
inline bool mc_index_load(McHostIndex &X, uint64_t input_hash, int nseq, const char *path)
{
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    McIxcHeader H;
    bool ok = fread(&H, sizeof H, 1, f) == 1 && H.magic == MC_IXC_MAGIC && H.version == MC_IXC_VERSION && H.input_hash == input_hash && (int)H.nseq == nseq;
    if (ok) {   // the sizes must add up to the file's
        uint64_t want = sizeof H + H.names_bytes + 8;
        const size_t es[12] = {1, 1, 4, 4, 4, 2, 4, sizeof(McBucketRec), 4, 4, 4, 8};
        for (int i = 0; i < 12; i++) { if (H.n[i] > (1ull << 32)) ok = false; want += H.n[i] * es[i]; }
        struct stat sb;
        ok = ok && fstat(fileno(f), &sb) == 0 && (uint64_t)sb.st_size == want && H.names_bytes < (1ull << 30);
    }
    uint64_t sum = 0;
    std::string names;
    if (ok) {
        sum = mc_ixc_sum(&H, sizeof H, 0);
        names.resize((size_t)H.names_bytes);
        ok = names.empty() || fread(&names[0], 1, names.size(), f) == names.size();
        sum = mc_ixc_sum(names.data(), names.size(), sum);
    }
#define MC_F(i, v) if (ok) { (v).resize((size_t)H.n[i]); const size_t b_ = (v).size() * sizeof((v)[0]); ok = b_ == 0 || fread((v).data(), 1, b_, f) == b_; sum = mc_ixc_sum((v).data(), b_, sum); }
    MC_IXC_ARRAYS(X, MC_F)
#undef MC_F
    uint64_t stored = 0;
    ok = ok && fread(&stored, 8, 1, f) == 1 && stored == sum;
    fclose(f);
    if (!ok) return false;
    X.names.clear();
    for (size_t at = 0; at < names.size();) { const size_t l = strlen(names.c_str() + at); X.names.emplace_back(names.c_str() + at, l); at += l + 1; }
    if ((int)X.names.size() != nseq) return false;
    X.nseq = nseq; X.rt_mask = H.rt_mask; X.max_bucket = H.max_bucket; X.freq_thr = H.freq_thr; X.nres = H.nres;
    for (int g = 0; g < 10; g++) X.letter_p[g] = H.letter_p[g];
    return true;
}

// A loaded index is only as good as the file it came from: a hash collision of the 64-bit input hash, a forgotten MC_IXC_VERSION bump
// or a foreign file in MC_INDEX_CACHE=<dir> would hand the kernels arrays they index unchecked.  So after a load the
// names and residues are compared with what the caller gave (cheap: 3.7 MB), and every offset the kernels follow is range-checked:
// off[] ascending to nres, bstart[] ascending to post.size(), every posting inside its sequence, every bucket record inside its bucket,
// the sizes of the fixed-size structures.  Anything off: the file is ignored and the index rebuilt.
inline bool mc_index_matches_input(const McHostIndex &X, const char *const *names, const char *const *seqs, int nseq)
{
    if (X.nseq != nseq || (int)X.names.size() != nseq || X.off.size() != (size_t)nseq + 1 || X.off[0] != 0) return false;
    if (X.res.size() != (size_t)X.nres || X.res_code.size() != X.res.size() || X.off[(size_t)nseq] != (uint32_t)X.nres) return false;
    for (int s = 0; s < nseq; s++) {
        if (X.names[(size_t)s] != names[s]) return false;
        const size_t l = strlen(seqs[s]);
        if (l >= 2048 || X.off[(size_t)s] > X.off[(size_t)s + 1] || X.off[(size_t)s + 1] - X.off[(size_t)s] != (uint32_t)l) return false;
        const uint8_t *r = &X.res[X.off[(size_t)s]], *rc = &X.res_code[X.off[(size_t)s]];
        for (size_t k = 0; k < l; k++) if (r[k] != (uint8_t)mc_dense_of_char((unsigned char)seqs[s][k]) || rc[k] != (uint8_t)mc_code_of_char((unsigned char)seqs[s][k])) return false;
    }
    if (X.bstart.size() != (size_t)MC_NBUCKET + 1 || X.bstart[0] != 0 || X.bstart[MC_NBUCKET] != X.post.size()) return false;
    if (X.keys.size() != X.post.size() + 64 || X.bitmap.size() != (size_t)(MC_NBUCKET + 31) / 32) return false;
    uint32_t maxb = 0;
    for (int b = 0; b < MC_NBUCKET; b++) {
        if (X.bstart[(size_t)b] > X.bstart[(size_t)b + 1]) return false;
        const uint32_t n = X.bstart[(size_t)b + 1] - X.bstart[(size_t)b];
        if (n > maxb) maxb = n;
        if ((((X.bitmap[(size_t)b >> 5] >> (b & 31)) & 1u) != 0) != (n != 0)) return false;
    }
    if (maxb != X.max_bucket) return false;
    for (size_t i = 0; i < X.post.size(); i++) {
        const uint32_t sq = X.post[i] >> 11, pos = X.post[i] & 0x7ffu;
        if (sq >= (uint32_t)nseq || pos + 6 >= X.off[(size_t)sq + 1] - X.off[(size_t)sq]) return false;
    }
    if (!X.rec.empty()) {
        if (X.rec.size() != (size_t)MC_NBUCKET) return false;
        for (int b = 0; b < MC_NBUCKET; b++) {
            const McBucketRec &R = X.rec[(size_t)b];
            if (R.start != X.bstart[(size_t)b] || R.cum[0] != 0 || R.cum[11] != X.bstart[(size_t)b + 1] - X.bstart[(size_t)b]) return false;
            for (int k = 0; k < 11; k++) if (R.cum[k] > R.cum[k + 1]) return false;
        }
    }
    if (X.filt.size() != (size_t)MC_FILT9_WORDS || X.wild.size() != (size_t)MC_WILD_LINES * 8 || X.pair.size() != (size_t)MC_PAIR_BLOCKS * 4) return false;
    if (X.rt.size() != (size_t)X.rt_mask + 1 || (X.rt_mask & (X.rt_mask + 1)) != 0) return false;
    return true;
}

// Proof obligation of mc_seg_mask_fx: for every composition a window of W residues can have (every partition of every
// t <= W), the integer tests must decide like Seg::entropy_cal's doubles.  Returns the number of disagreements (0).
inline int mc_seg_fx_verify(const McTables &T, double *min_margin)
{
    int bad = 0;
    double mm = 1e9;
    for (int w = 0; w < 2; w++) {
        const int W = w ? 8 : 12;
        uint8_t sv[24];
        // enumerate partitions of t into parts in non-increasing order
        for (int t = 0; t <= W; t++) {
            int parts[16], np = 0;
            // iterative partition enumeration
            std::vector<std::vector<int>> all;
            std::vector<int> cur;
            struct Rec { static void go(int rem, int maxp, std::vector<int> &cur, std::vector<std::vector<int>> &all) {
                if (rem == 0) { all.push_back(cur); return; }
                for (int p = std::min(rem, maxp); p >= 1; p--) { cur.push_back(p); go(rem - p, p, cur, all); cur.pop_back(); } } };
            Rec::go(t, t, cur, all);
            (void)parts; (void)np;
            for (const auto &pt : all) {
                if (pt.size() > 20) continue;                  // only 20 residue classes exist
                int S = 0;
                for (size_t i = 0; i < pt.size(); i++) { sv[i] = (uint8_t)pt[i]; int c = 0; for (int k = 0; k < pt[i]; k++) { S += T.seg_din[c]; c++; } }
                sv[pt.size()] = 0;
                const double H = mc_seg_entropy(T, W, sv);
                const bool lo = S >= T.seg_tlo[t], hi = S >= T.seg_thi[t];
                if (lo != (H <= 2.2) || hi != (H <= 2.5)) bad++;
                mm = std::min(mm, std::min(fabs(H - 2.2), fabs(H - 2.5)));
            }
        }
    }
    if (min_margin) *min_margin = mm;
    return bad;
}

// ---- statistics / constant tables (BlastStat::*@0x437d50-0x438b40, Seg::initialize@0x439650) ----------
inline int mc_length_adjustment(double n, double nseq, int qlen)
{ // BlastStat::blastComputeLengthAdjustment@0x438410 (NCBI BLAST_ComputeLengthAdjustment, gapped set)
    const double K = 0.041, beta = -30.0;
    union { uint64_t u; double d; } ad; ad.u = 0x401c76e43aa79bbbULL;
    double alpha_d_lambda = ad.d, logK = log(K), m = (double)qlen;
    double ell, ell_min = 0.0, ell_max, ell_next = 0.0, ss, ell_bar, mb, c, mx = (m > n) ? m : n;
    bool converged = false;
    c = m * n - mx / K;
    if (c < 0) return 0;
    mb = m * nseq + n;
    ell_max = 2 * c / (mb + sqrt(mb * mb + (-4.0) * nseq * c));
    for (int i = 1; i <= 20; i++) {
        ell = ell_next;
        ss = (m - ell) * (n - nseq * ell);
        ell_bar = alpha_d_lambda * (log(ss) + logK) + beta;
        if (ell_bar >= ell) { ell_min = ell; if (ell_bar - ell_min <= 1.0) { converged = true; break; } if (ell_min == ell_max) break; }
        else ell_max = ell;
        if (ell_min <= ell_bar && ell_bar <= ell_max) ell_next = ell_bar;
        else ell_next = (i == 1) ? ell_max : (ell_min + ell_max) * 0.5;
    }
    int adj = (int)ell_min;
    if (converged) {
        ell = ceil(ell_min);
        if (ell <= ell_max) { ss = (m - ell) * (n - nseq * ell); if (alpha_d_lambda * (log(ss) + logK) + beta >= ell) adj = (int)ell; }
    }
    return adj;
}

// mc_segtab_*: every (length <= 15, state vector) pair -> order-preserving key of mc_rg_getprob.  State vectors are the
// non-increasing sequences of counts with sum t <= length (residues of the window that are not amino acids count for the
// length only).
inline void mc_build_segtab(const double *lnfac, std::vector<uint64_t> &tab)
{
    tab.assign((size_t)MC_SEGTAB_SLOTS * 2, 0);
    size_t npairs = 0;
    // enumerate the partitions recursively: parts[0] >= parts[1] >= ... > 0
    struct Rec {
        const double *lnfac; std::vector<uint64_t> &tab; size_t &npairs;
        void emit(uint64_t sv, int t)
        {
            for (int len = (t > 2 ? t : 2); len <= 15; len++) {
                const uint64_t k = mc_rh_of_sv(sv) | ((uint64_t)len << 60);
                uint64_t ck = k, cv = mc_seg_prob_key(mc_rg_getprob(lnfac, sv, len));
                // cuckoo placement: the pair goes to its first slot and whoever sat there moves to its other one (2,452 pairs, 8,192 slots)
                uint32_t at = 0xFFFFFFFFu;
                for (int kick = 0;; kick++) {
                    if (kick >= 512) { fprintf(stderr, "mc_build_segtab: no placement\n"); abort(); }   // (not reached at this load; deterministic)
                    uint32_t h1, h2;
                    mc_segtab_slots(ck, h1, h2);
                    const uint32_t h = (at == h1) ? h2 : h1;                 // a displaced pair takes the slot it did not come from
                    std::swap(ck, tab[2 * h]); std::swap(cv, tab[2 * h + 1]);
                    at = h;
                    if (ck == 0) break;
                }
                npairs++;
            }
        }
        void go(uint64_t sv, int nparts, int t, int maxpart)
        {
            emit(sv, t);
            if (nparts == 15) return;
            for (int p = 1; p <= maxpart && t + p <= 15; p++) go(sv | ((uint64_t)p << (4 * nparts)), nparts + 1, t + p, p);
        }
    } rec{lnfac, tab, npairs};
    rec.go(0, 0, 0, 15);
    (void)npairs;
}

inline void mc_fill_tables(McTables &T, const McHostIndex &X, int read_len, double loge_thr)
{
    static const int8_t B62[20][20] = {
        {4, -1, -2, -2, 0, -1, -1, 0, -2, -1, -1, -1, -1, -2, -1, 1, 0, -3, -2, 0}, {-1, 5, 0, -2, -3, 1, 0, -2, 0, -3, -2, 2, -1, -3, -2, -1, -1, -3, -2, -3},
        {-2, 0, 6, 1, -3, 0, 0, 0, 1, -3, -3, 0, -2, -3, -2, 1, 0, -4, -2, -3}, {-2, -2, 1, 6, -3, 0, 2, -1, -1, -3, -4, -1, -3, -3, -1, 0, -1, -4, -3, -3},
        {0, -3, -3, -3, 9, -3, -4, -3, -3, -1, -1, -3, -1, -2, -3, -1, -1, -2, -2, -1}, {-1, 1, 0, 0, -3, 5, 2, -2, 0, -3, -2, 1, 0, -3, -1, 0, -1, -2, -1, -2},
        {-1, 0, 0, 2, -4, 2, 5, -2, 0, -3, -3, 1, -2, -3, -1, 0, -1, -3, -2, -2}, {0, -2, 0, -1, -3, -2, -2, 6, -2, -4, -4, -2, -3, -3, -2, 0, -2, -2, -3, -3},
        {-2, 0, 1, -1, -3, 0, 0, -2, 8, -3, -3, -1, -2, -1, -2, -1, -2, -2, 2, -3}, {-1, -3, -3, -3, -1, -3, -3, -4, -3, 4, 2, -3, 1, 0, -3, -2, -1, -3, -1, 3},
        {-1, -2, -3, -4, -1, -2, -3, -4, -3, 2, 4, -2, 2, 0, -3, -2, -1, -2, -1, 1}, {-1, 2, 0, -1, -3, 1, 1, -2, -1, -3, -2, 5, -1, -3, -1, 0, -1, -3, -2, -2},
        {-1, -1, -2, -3, -1, 0, -2, -3, -2, 1, 2, -1, 5, 0, -2, -1, -1, -1, -1, 1}, {-2, -3, -3, -3, -2, -3, -3, -3, -1, 0, 0, -3, 0, 6, -4, -2, -2, 1, 3, -1},
        {-1, -2, -2, -1, -3, -1, -1, -2, -2, -3, -3, -1, -2, -4, 7, -1, -1, -4, -3, -2}, {1, -1, 1, 0, -1, 0, 0, 0, -1, -2, -2, 0, -1, -2, -1, 4, 1, -3, -2, -2},
        {0, -1, 0, -1, -1, -1, -1, -2, -2, -1, -1, -1, -1, -2, -1, 1, 5, -2, -2, 0}, {-3, -3, -4, -4, -2, -2, -3, -2, -2, -3, -2, -3, -1, 1, -4, -3, -2, 11, 2, -3},
        {-2, -2, -2, -3, -2, -1, -2, -3, 2, -1, -1, -2, -1, 3, -3, -2, -2, 2, 7, -1}, {0, -3, -3, -3, -1, -2, -2, -3, -3, 3, 1, -2, 1, -1, -2, -2, 0, -3, -1, 4}};
    static const char *aa = "FFLLSSSSYY..CC.WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
    const double LN2 = 0.6931471805599453;
    memset(&T, 0, sizeof T);
    for (int a = 0; a < 32; a++) for (int b = 0; b < 32; b++) T.sub[(a << 5) | b] = (a < 20 && b < 20) ? B62[a][b] : -5;
    for (int d = 0; d < 32; d++) T.grp[d] = (uint8_t)mc_group_of_dense(d);
    for (int i = 0; i < 64; i++) T.codon[i] = (uint8_t)mc_dense_of_char((unsigned char)aa[i]);
    for (int w = 0; w < 2; w++) { int W = w ? 8 : 12; T.entray[w][0] = 0.0; for (int i = 1; i <= W; i++) { double p = (double)i / (double)W; T.entray[w][i] = (-p) * log(p) / LN2; } }
    for (int tot = 1; tot <= 12; tot++) { double inv = 1.0 / (double)tot; for (int c = 1; c <= tot; c++) { double x = (double)c; T.lterm[tot][c] = log(inv * x) * x; } }
    { char buf[64]; for (int i = 0; i < MC_LNFAC_N; i++) { snprintf(buf, sizeof buf, "%.6f", lgamma((double)i + 1.0)); T.lnfac[i] = atof(buf); } }
    {   // fixed-point window-entropy tests of mc_seg_mask_fx
        const double sc = (double)(1 << MC_SEG_FXBITS);
        int32_t F[16]; memset(F, 0, sizeof F);
        for (int c = 2; c <= 12; c++) F[c] = (int32_t)llround((double)c * log2((double)c) * sc);
        for (int c = 0; c < 16; c++) {
            T.seg_dout[c] = (c >= 1 && c <= 12) ? F[c - 1] - F[c] : 0;
            T.seg_din[c] = (c <= 11) ? F[c + 1] - F[c] : 0;
            T.seg_tlo[c] = T.seg_thi[c] = 0x7fffffff;
        }
        T.seg_tlo[0] = T.seg_thi[0] = -0x7fffffff;             // an empty window has entropy 0: below both cuts
        for (int t = 1; t <= 12; t++) {
            T.seg_tlo[t] = (int32_t)llround((double)t * (log2((double)t) - 2.2) * sc);
            T.seg_thi[t] = (int32_t)llround((double)t * (log2((double)t) - 2.5) * sc);
        }
    }
    T.gap_trigger = (25.0 * LN2 - 2.0099154790312257) / 0.318;
    T.xdrop_ungapped = (7.0 * LN2 - 2.0099154790312257) / 0.318;
    T.xdrop_gapped = (15.0 * LN2 - 3.1941832122778293) / 0.267;
    // (the kernels test `best - score > xdrop` on integers as `best - score >= floor(xdrop) + 1`: exact unless a threshold sits on an
    // integer, where the reference's double arithmetic would decide - 8.94 and 26.98 are far from one; a change of the constants that
    // moves them there must fail loudly, not drift)
    for (double x : {T.xdrop_ungapped, T.xdrop_gapped}) if (fabs(x - nearbyint(x)) < 1e-6) { fprintf(stderr, "mc_fill_tables: an X-drop threshold on an integer (%.9f): the integer exit tests of the kernels are not exact there\n", x); abort(); }
    T.loge_thr = loge_thr;
    T.freq_thr = X.freq_thr;
    for (int g = 0; g < 10; g++) T.letter_p[g] = X.letter_p[g];
    int m = read_len / 3;
    int adj = (m <= 10) ? 0 : mc_length_adjustment((double)X.nres, (double)X.nseq, m);
    T.ell = (double)adj; T.mprime = (double)m - T.ell; T.nprime = (double)X.nres - (double)X.nseq * T.ell; T.logK = log(0.041);
    for (int s = 0; s < MC_SMAX; s++) {
        double t = 0.041 * T.nprime; t = t * T.mprime;
        double x = exp((double)s * (-0.267)) * t; x = x / (1.0 - 0.1);
        double le = (x == 0.0) ? -10000.0 : log(x) / 2.302585092994046;
        le = (le > 0.0) ? floor(le * 100.0 + 0.5) / 100.0 : floor(le * 100.0 - 0.5) / 100.0;
        T.loge_r[s] = le;
        double bits = ((double)s * 0.267 - T.logK) / LN2;
        T.bits_r[s] = floor(bits * 100.0 + 0.5) / 100.0;
    }
}
