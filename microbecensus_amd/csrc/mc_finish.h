// mc_finish.h - per-read ranking / linking / capping / classification: what ONE GPU thread does for one
// read that produced HSPs.  Same conventions as mc_core.h (MC_HD code shared with the test-only emulation).
//
// Follows, for the read's HSPs: the multimap bookkeeping of CalRes (0x4082b0-0x408446: duplicate test
// against the newest HSP of the same subject, otherwise "insert in front"), PrintRes@0x409310
// (SumEvalue@0x408a50 per subject, std::sort by log E, 500-row cap, log E < threshold), the heap
// permutation MergeRes@0x40e3b0 applies to rows whose printed log E is equal, and then MicrobeCensus'
// classify_reads / alignment_filter / alignment_coverage (/root/reference/microbe_census/
// microbe_census.py:400-460) on the surviving rows.
#pragma once
#include "mc_core.h"
#include <math.h>

struct McClassPars {               // find_opt_pars(pars.map, L)  microbe_census.py:61-72
    double min_cov[32], min_score[32];
    int32_t max_aaid[32];          // always an integer in pars.map: pid > max_aaid  <=>  100*nmatch > max_aaid*alnlen
    int32_t aln_stat[32];          // 0 hits, 1 cov, 2 aln
    int32_t nfam, read_len;
};

struct McBestHit {                 // best_hits[read] = [fam, aln, aln/target_len, score]  microbe_census.py:450-453
    int32_t read, family, aln, target_len;
    double bits;
};

// ------------------------------------------------------------------------------------------------
// libstdc++ (GCC 4.4) std::sort on McHsp with a '<' key, exactly as the binary instantiates it
// (__introsort_loop@0x42a570/0x42a0b0/0x42a310, __final_insertion_sort@0x4263f0, heap fallback).
// key selector: 0 = loge, 1 = frame, 2 = qaas
// ------------------------------------------------------------------------------------------------
MC_HD bool mc_hless(const McHsp &a, const McHsp &b, int key)
{
    return key == 0 ? (a.loge < b.loge) : key == 1 ? (a.frame < b.frame) : (a.qaas < b.qaas);
}
// The same sort on 16-byte (key, index) items: the sequence of comparisons and moves of std::sort depends on the keys
// only, so sorting the items and reading the records through the indices gives the permutation the reference gets by
// sorting the 48-byte records themselves - at a third of the memory traffic (the final by-log-E sort of a read).
struct McSortItem { double k; uint32_t i; uint32_t pad; };
MC_HD bool mc_hless(const McSortItem &a, const McSortItem &b, int) { return a.k < b.k; }

#define MC_SORT_FN(x) x
#define MC_SORT_ATTR MC_HDN
#include "mc_sort_impl.h"
#undef MC_SORT_FN
#undef MC_SORT_ATTR
#define MC_SORT_FN(x) x##_inl
#define MC_SORT_ATTR MC_HD
#include "mc_sort_impl.h"
#undef MC_SORT_FN
#undef MC_SORT_ATTR
MC_HDN void mc_stable_sort_loge(McHsp *first, long n)
{ // std::stable_sort(CompEvalueObj): any stable sort produces the same permutation
    for (long i = 1; i < n; ++i) {
        McHsp val = first[i]; long j = i;
        while (j > 0 && val.loge < first[j - 1].loge) { first[j] = first[j - 1]; --j; }
        first[j] = val;
    }
}

#if defined(MC_EXP_TIMING) && defined(__HIPCC__)
__device__ unsigned long long g_fr_acc2[8];          // timing build: inside the groups - 0 stack 1 sort by frame 2 sort by start + stable sort 3 choice 4 sum statistics 5 copy back
#endif
#if defined(MC_EXP_TIMING) && defined(__HIP_DEVICE_COMPILE__)
#define MC_FG_BEGIN unsigned long long fg_last_ = __builtin_readcyclecounter()
#define MC_FG_TICK(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); atomicAdd(&g_fr_acc2[k], now_ - fg_last_); fg_last_ = now_; } while (0)
#else
#define MC_FG_BEGIN do { } while (0)
#define MC_FG_TICK(k) do { } while (0)
#endif
// ------------------------------------------------------------------------------------------------
// sum statistics (BlastStat::sumScore2Expect@0x438300 -> @0x437f90)
// ------------------------------------------------------------------------------------------------
MC_HD double mc_fac(int n) { int r = 1; while (n > 1) { r *= n; n--; } return (double)r; }
MC_HDN double mc_sum_expect(const McTables &T, int n, const double *scores, int subj_len)
{
    double sum = 0.0;
    for (int i = 0; i < n; i++) sum = sum + scores[i];
    double a = 1.0 / 0.041, b = (double)subj_len - T.ell;
    if (!(a > b)) a = b;
    double t = log(0.041 * T.mprime * a);
    double xsum = sum * 0.267 - t;
    xsum = xsum - (double)(n - 1) * (T.logK + 7.824046010856292);
    xsum = xsum - log(mc_fac(n));
    double ex = exp(-xsum);
    double pw = pow(xsum, (double)(n - 1));
    double d = pow(0.1, (double)(n - 1)) * 0.9;
    double r = T.nprime / (double)subj_len;
    double x = ex * pw;
    x = x / (mc_fac(n) * mc_fac(n - 1));
    x = x / d;
    return r * x;
}

// CHashSearch::SumEvalue@0x408a50 on v[st, ed); returns the new end (the range may shrink).
// tmp must hold 2*(ed-st) entries.
// The reference sorts the subject's HSPs three times (std::sort by frame, per strand std::sort by start and std::stable_sort by
// log E) and picks a consistent chain.  All of that depends on five small integers per HSP, so it runs on one 64-bit word per HSP
// (frame | start | end | score | position) - the sequence of comparisons and moves of the sorts depends on the keys only - and the
// 48-byte records are moved once at the end (a thread walks global memory alone here: sorting the records themselves was the
// thread-per-read kernel's time).  log E of a single HSP is the table value of its score: a < b in log E <=> a > b in score.
struct McLinkItem { uint64_t w; };
#define MC_LK_IDX(x) ((int)((x).w & 0xFFFFFu))                 // position in the subject's stack (20 bits)
#define MC_LK_SCORE(x) ((int)(((x).w >> 20) & 0xFFFFu))
#define MC_LK_QAAE(x) ((int)(((x).w >> 36) & 0xFFu))
#define MC_LK_QAAS(x) ((int)(((x).w >> 44) & 0xFFu))
#define MC_LK_FRAME(x) ((int)(((x).w >> 52) & 0xFu))             // (bits 62, 63: the HSP carries the log E of its strand's chain)
MC_HD bool mc_hless(const McLinkItem &a, const McLinkItem &b, int key) { return key == 1 ? MC_LK_FRAME(a) < MC_LK_FRAME(b) : MC_LK_QAAS(a) < MC_LK_QAAS(b); }
MC_HDN int mc_sum_evalue(const McTables &T, McHsp *v, int st, int ed, int subj_len, McHsp *__restrict__ tmp)
{
    McHsp *__restrict__ a = v + st;                                 // (v and tmp never overlap: said so, or every copy below waits for the store before it)
    const int n = ed - st;
    int part, nres = 0;
    MC_FG_BEGIN;
    // tmp: n records (the copy the result is gathered from), behind them three arrays of n words
    McLinkItem *it = (McLinkItem *)(tmp + n), *res = it + n, *chosen = res + n;
    for (int i = 0; i < n; i++) it[i].w = ((uint64_t)((uint16_t)a[i].frame & 0xFu) << 52) | ((uint64_t)(uint8_t)a[i].qaas << 44) | ((uint64_t)(uint8_t)a[i].qaae << 36) | ((uint64_t)(uint16_t)a[i].score << 20) | (uint64_t)i;
    mc_std_sort(it, n, 1);
    MC_FG_TICK(1);
    for (part = 0; part < n && !(MC_LK_FRAME(it[part]) > 2); part++) {}
    double le_pass[2] = {0.0, 0.0};
    const bool link = !((n - part) <= 1 && part <= 1);
    if (link)
        for (int pass = 0; pass < 2; pass++) {
            McLinkItem *g = pass ? it + part : it;
            int gn = pass ? n - part : part, nc = 0;
            if (gn == 0) continue;
            if (gn == 1) { if (T.loge_thr > T.loge_r[MC_LK_SCORE(g[0])]) res[nres++] = g[0]; continue; }
            MC_FG_TICK(5);
            mc_std_sort(g, gn, 2);
            // std::stable_sort(CompEvalueObj): any stable sort produces the same permutation.  Insertion for the short strands; a bottom-up
            // merge sort (scratch: `chosen`, not in use yet) for the long ones - a subject with 150 HSPs on a strand is 5,600 dependent moves
            // of an insertion sort for the one lane of k_finish_heavy that has it (round 5)
            if (gn <= 12) {
                for (long i = 1; i < gn; ++i) {
                    const McLinkItem val = g[i]; long j = i;
                    while (j > 0 && MC_LK_SCORE(val) > MC_LK_SCORE(g[j - 1])) { g[j] = g[j - 1]; --j; }
                    g[j] = val;
                }
            } else {
                for (int w = 1; w < gn; w *= 2) {
                    for (int lo = 0; lo < gn; lo += 2 * w) {
                        const int mid = lo + w < gn ? lo + w : gn, hi = lo + 2 * w < gn ? lo + 2 * w : gn;
                        int x = lo, y = mid, o = lo;
                        while (x < mid && y < hi) { if (MC_LK_SCORE(g[y]) > MC_LK_SCORE(g[x])) chosen[o++] = g[y++]; else chosen[o++] = g[x++]; }   // (on a tie the left one first: stable)
                        while (x < mid) chosen[o++] = g[x++];
                        while (y < hi) chosen[o++] = g[y++];
                    }
                    for (int i = 0; i < gn; i++) g[i] = chosen[i];
                }
            }
            MC_FG_TICK(2);
            chosen[nc++] = g[0];
            for (int i = 1; i < gn; i++) {
                const McLinkItem e = g[i];
                const int es = MC_LK_QAAS(e), ee = MC_LK_QAAE(e);
                int ov = (ee + 1 - es) >> 1;
                bool ok = true;
                if (ov > 10) ov = 10;
                if (T.loge_r[MC_LK_SCORE(e)] >= 1.0 && !(MC_LK_SCORE(e) > 30)) continue;
                for (int j = 0; j < nc; j++) {
                    const int cs = MC_LK_QAAS(chosen[j]), ce = MC_LK_QAAE(chosen[j]);
                    if (es <= ce - ov) { if (ee >= cs + ov) { ok = false; break; } }
                    if (ee - ov < cs) continue;
                    if (ce >= ov + es) { ok = false; break; }
                }
                if (ok) chosen[nc++] = e;
            }
            MC_FG_TICK(3);
            if (nc == 1) { if (T.loge_thr > T.loge_r[MC_LK_SCORE(chosen[0])]) res[nres++] = chosen[0]; }
            else {
                double sc[5];
                int k = nc < 5 ? nc : 5;
                for (int i = 0; i < k; i++) sc[i] = (double)MC_LK_SCORE(chosen[i]);
                double E = mc_sum_expect(T, k, sc, subj_len);
                double le = (E == 0.0) ? -10000.0 : log(E) / 2.302585092994046;
                le_pass[pass] = le;
                if (T.loge_thr > le) for (int i = 0; i < nc; i++) { res[nres].w = chosen[i].w | (1ull << (62 + pass)); nres++; }   // (the chain's members carry its log E)
            }
            MC_FG_TICK(4);
        }
    // the records: the chains that stay, or - nothing stays - all of them in the order the sorts left them in
    const McLinkItem *out = nres > 0 ? res : it;
    const int nout = nres > 0 ? nres : n;
    bool moved = nout != n;
    for (int i = 0; i < n && !moved; i++) moved = MC_LK_IDX(out[i]) != i || (out[i].w >> 62) != 0;
    if (moved) {
        int i = 0;
        for (; i + 2 <= n; i += 2) { const McHsp r0 = a[i], r1 = a[i + 1]; tmp[i] = r0; tmp[i + 1] = r1; }   // (two records' loads in flight per turn)
        for (; i < n; i++) tmp[i] = a[i];
        for (i = 0; i + 2 <= nout; i += 2) {
            const uint64_t w0 = out[i].w, w1 = out[i + 1].w;
            McHsp h0 = tmp[w0 & 0xFFFFFu], h1 = tmp[w1 & 0xFFFFFu];
            if (w0 >> 62) h0.loge = le_pass[(w0 >> 63) & 1];
            if (w1 >> 62) h1.loge = le_pass[(w1 >> 63) & 1];
            a[i] = h0; a[i + 1] = h1;
        }
        for (; i < nout; i++) {
            McHsp h = tmp[MC_LK_IDX(out[i])];
            if (out[i].w >> 62) h.loge = le_pass[(out[i].w >> 63) & 1];
            a[i] = h;
        }
    }
    MC_FG_TICK(5);
    return nres > 0 ? st + nres : ed;
}

// ------------------------------------------------------------------------------------------------
// printed-log-E key: two rows tie in MergeRes when "%g" prints the same 6 significant digits
// ------------------------------------------------------------------------------------------------
MC_HDN double mc_round6(double x)
{
    const double p10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    if (x == 0.0 || x != x) return x;
    double ax = x < 0 ? -x : x;
    int e = 0;
    if (ax >= 1.0) { while (e < 16 && ax >= p10[e + 1]) e++; }
    else { while (e > -16 && ax < 1.0 / p10[-e]) e--; }
    double scaled = (5 - e >= 0) ? ax * p10[5 - e] : ax / p10[e - 5];
    double r = nearbyint(scaled);
    if (r >= 1000000.0) { r = r / 10.0; e++; }
    r = (5 - e >= 0) ? r / p10[5 - e] : r * p10[e - 5];
    return x < 0 ? -r : r;
}
MC_HDN void mc_row_adjust_heap(McRow *a, double *k, long hole, long len, McRow v, double kv)
{
    long top = hole, sc = hole;
    while (sc < (len - 1) / 2) {
        sc = 2 * (sc + 1);
        if (k[sc] < k[sc - 1]) sc--;
        a[hole] = a[sc]; k[hole] = k[sc]; hole = sc;
    }
    if ((len & 1) == 0 && sc == (len - 2) / 2) { sc = 2 * (sc + 1); a[hole] = a[sc - 1]; k[hole] = k[sc - 1]; hole = sc - 1; }
    long parent = (hole - 1) / 2;
    while (hole > top && k[parent] < kv) { a[hole] = a[parent]; k[hole] = k[parent]; hole = parent; parent = (hole - 1) / 2; }
    a[hole] = v; k[hole] = kv;
}
// std::partial_sort(first, first+n, first+n) keyed by the printed log E (MergeRes 0x40ed5e-0x40f007)
MC_HDN void mc_merge_res_order(McRow *a, double *k, int n)
{
    if (n < 2) return;
    for (long i = (n - 2) / 2;; i--) { mc_row_adjust_heap(a, k, i, n, a[i], k[i]); if (i == 0) break; }
    for (long m = n; m > 1;) { m--; McRow v = a[m]; double kv = k[m]; a[m] = a[0]; k[m] = k[0]; mc_row_adjust_heap(a, k, 0, m, v, kv); }
}

// ------------------------------------------------------------------------------------------------
// alignment_coverage + alignment_filter (microbe_census.py:400-430) on one row, all in IEEE double
// ------------------------------------------------------------------------------------------------
// alignment_coverage (microbe_census.py:400-418; training/training.py:248-265 computes the same): aln / the longest alignment
// the read could have had at this position of the target
MC_HD double mc_row_coverage(int read_len, const McRow &r, int target_len)
{
    double query_len = (double)read_len / 3.0;
    double qs = (double)(r.qstart < r.qend ? r.qstart : r.qend), qe = (double)(r.qstart < r.qend ? r.qend : r.qstart);
    double md = fmod(qs, 3.0);
    double frame = (md == 1.0 || md == 2.0) ? md : 3.0;
    double query_start = (qs + 3.0 - frame) / 3.0;
    double query_stop = (qe + 1.0 - frame) / 3.0;
    double t1 = (double)r.sstart + 1.0, t2 = (double)r.send + 1.0;
    double t_start = t1 < t2 ? t1 : t2, t_stop = t1 < t2 ? t2 : t1;
    double x1 = query_start - 1.0, x2 = t_start - 1.0;
    double x = x1 < x2 ? x1 : x2;       // python min(a, b): b if b < a else a
    double y = (double)r.alnlen;
    double z1 = query_len - query_stop, z2 = (double)target_len - t_stop;
    double z = z2 < z1 ? z2 : z1;
    double maxaln = x + y + z;
    return (double)r.alnlen / maxaln;
}
MC_HD bool mc_row_passes(const McClassPars &P, const McRow &r, int fam, int target_len, int nmatch)
{
    double query_len = (double)P.read_len / 3.0;
    double qs = (double)(r.qstart < r.qend ? r.qstart : r.qend), qe = (double)(r.qstart < r.qend ? r.qend : r.qstart);
    double md = fmod(qs, 3.0);
    double frame = (md == 1.0 || md == 2.0) ? md : 3.0;
    double query_start = (qs + 3.0 - frame) / 3.0;
    double query_stop = (qe + 1.0 - frame) / 3.0;
    double t1 = (double)r.sstart + 1.0, t2 = (double)r.send + 1.0;
    double t_start = t1 < t2 ? t1 : t2, t_stop = t1 < t2 ? t2 : t1;
    double x1 = query_start - 1.0, x2 = t_start - 1.0;
    double x = x1 < x2 ? x1 : x2;       // python min(a, b): b if b < a else a
    double y = (double)r.alnlen;
    double z1 = query_len - query_stop, z2 = (double)target_len - t_stop;
    double z = z2 < z1 ? z2 : z1;
    double maxaln = x + y + z;
    double cov = (double)r.alnlen / maxaln;
    if (cov < P.min_cov[fam]) return false;
    if (r.bits < P.min_score[fam]) return false;
    if (100 * nmatch > P.max_aaid[fam] * r.alnlen) return false;
    return true;
}

// ------------------------------------------------------------------------------------------------
// One read.  in[0,n): its HSPs sorted by (subject, chrono).  v: n entries, tmp: 2n entries of scratch.
// rows_out (may be null) receives the m8 rows (<= 500) in the order the reference prints them;
// krows: 500 doubles of scratch.  Returns the number of rows; *best gets the classify_reads result
// (best->family = -1 when no row passes the filters).
// ------------------------------------------------------------------------------------------------
// HSPs in[a, b) of one subject (sorted by chrono) -> out[0, kept): the multimap's view of them, linked by sum statistics.
// tmp: 2 (b - a) entries of scratch.
MC_HDN int mc_group_stack(const McHsp *in, int a, int b, McHsp *out)
{
    int vn = 0;
    // stack of this subject's HSPs, newest on top; a re-found HSP only replaces the top if it is better.  The top of the stack is
    // kept in registers and written when the next HSP goes on top of it (the thread walks global memory alone: reading back what
    // it has just written was a trip per HSP); the record behind the current one is read ahead.
    McHsp t = in[a];
    McHsp nx = t;
    if (a + 1 < b) nx = in[a + 1];
    for (int k = a + 1; k < b; k++) {
        const McHsp h = nx;
        if (k + 1 < b) nx = in[k + 1];
        if (t.frame == h.frame && t.qaas == h.qaas && t.ds == h.ds && t.qaae == h.qaae && t.de == h.de) {
            if (t.loge > h.loge) { t.score = h.score; t.loge = h.loge; t.alnlen = h.alnlen; t.mism = h.mism; t.gaps = h.gaps; t.nmatch = h.nmatch; t.qnts = h.qnts; t.qnte = h.qnte; }
            continue;
        }
        out[vn++] = t;
        t = h;
    }
    out[vn++] = t;
#if defined(MC_EXP_TIMING) && defined(__HIP_DEVICE_COMPILE__)
    atomicAdd(&g_fr_acc2[6], 1ull); if (vn > 1) atomicAdd(&g_fr_acc2[7], 1ull);
#endif
    for (int i = 0, j = vn - 1; i < j; i++, j--) { McHsp t = out[i]; out[i] = out[j]; out[j] = t; }   // multimap order: newest first
    return vn;
}
MC_HDN int mc_finish_group(const McTables &T, const McIndex &X, const McHsp *in, int a, int b, McHsp *out, McHsp *tmp)
{
    const int sidx = in[a].sidx;
    int vn = mc_group_stack(in, a, b, out);
    if (vn > 1) vn = mc_sum_evalue(T, out, 0, vn, (int)(X.off[sidx + 1] - X.off[sidx]), tmp);
    return vn;
}
// one m8 row (PrintRes) from an HSP; the frame slot carries nmatch for the classifier
MC_HD void mc_fill_row(const McTables &T, int read_id, const McHsp &h, McRow &r)
{
    { uint32_t *z = (uint32_t *)&r; for (unsigned i = 0; i < sizeof(McRow) / 4; i++) z[i] = 0; }   // padding bytes too: the rows are handed out as raw memory
    r.query = read_id; r.subject = h.sidx; r.ident = (double)h.nmatch * 100.0 / (double)h.alnlen;
    r.alnlen = h.alnlen; r.mismatch = h.mism; r.gapopen = h.gaps; r.qstart = h.qnts; r.qend = h.qnte; r.sstart = h.ds; r.send = h.de;
    r.loge = h.loge; r.bits = T.bits_r[h.score]; r.score = h.score; r.frame = h.nmatch;
}

#if defined(MC_EXP_TIMING) && defined(__HIPCC__)
__device__ unsigned long long g_fr_acc[8];           // cycle counters of the timing build: wall time of a thread per phase, summed over the threads
#endif
#if defined(MC_EXP_TIMING) && defined(__HIP_DEVICE_COMPILE__)
#define MC_FR_BEGIN unsigned long long fr_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, fr_last_ = __builtin_readcyclecounter()
#define MC_FR_TICK(prev) do { const unsigned long long now_ = __builtin_readcyclecounter(); fr_acc_[prev] += now_ - fr_last_; fr_last_ = now_; } while (0)
#define MC_FR_END do { for (int k_ = 0; k_ < 8; k_++) if (fr_acc_[k_]) atomicAdd(&g_fr_acc[k_], fr_acc_[k_]); } while (0)
#else
#define MC_FR_BEGIN do { } while (0)
#define MC_FR_TICK(prev) do { } while (0)
#define MC_FR_END do { } while (0)
#endif
// The stacks of a read: in[0, n) sorted by (subject, chrono) -> v[0, vn): CalRes' view of every subject (mc_group_stack), subject by
// subject.  The first record of a subject's stack carries the stack's size in .read (the other records of the stack 0; neither .read
// nor .chrono is used again).  On the device the ordering kernels (k_order_*) build the same array from the sorted keys.
MC_HDN int mc_build_stacks(const McHsp *in, int n, McHsp *v)
{
    int vn = 0;
    McHsp cur = in[0];
    for (int a = 0; a < n;) {
        const bool lastone = a + 1 >= n;
        McHsp nxt = cur;
        if (!lastone) nxt = in[a + 1];
        if (lastone || nxt.sidx != cur.sidx) { cur.read = 1u; cur.chrono = 1u; v[vn++] = cur; cur = nxt; a++; continue; }   // a subject with ONE HSP (most subjects of most reads)
        int b = a + 2;
        const int sidx = cur.sidx;
        while (b < n && in[b].sidx == sidx) b++;
        const int k = mc_group_stack(in, a, b, v + vn);
        for (int j = 1; j < k; j++) v[vn + j].read = 0u;
        v[vn].read = (uint32_t)k; v[vn].chrono = (uint32_t)k;
        vn += k;
        a = b;
        if (a < n) cur = in[a];
    }
    return vn;
}
// One read from its stacks (v[0, vn) as mc_build_stacks leaves them; tmp: 2 vn entries of scratch): sum statistics per subject,
// std::sort by log E, the 500-row cap, MergeRes' order, the rows and the classification.
// (S: the two order-defining sorts - out of line, or forced inline for the kernel that keeps the items in LDS: inlining is what lets
// the compiler address them as LDS)
struct McSortsOut { static MC_HD void sort(McSortItem *a, long n) { mc_std_sort(a, n, 0); } static MC_HD void heap(McSortItem *a, long n) { mc_heapsort(a, n, 0); } };
struct McSortsInl { static MC_HD void sort(McSortItem *a, long n) { mc_std_sort_inl(a, n, 0); } static MC_HD void heap(McSortItem *a, long n) { mc_heapsort_inl(a, n, 0); } };
template <class S>
MC_HD int mc_finish_stacked_t(const McTables &T, const McIndex &X, const McClassPars &P, const int32_t *marker_family,
                              int read_id, McHsp *v, int vn, McHsp *tmp, McRow *rows, double *krows, McSortItem *items, McBestHit *best)
{
    MC_FR_BEGIN;
    // Sum statistics for the subjects with more than one HSP, in a loop of their own: the threads of a wave that have such a
    // subject link it AT THE SAME TIME - met one by one while the stacks were built, the wave went through the sorts and logarithms
    // of every one of them separately, which was the light kernel's time (cycle counters).  Then the kept HSPs are moved together,
    // if anything shrank.
    {
        bool shrank = false, any = false;
        for (int p = 0; p < vn;) {
            const int k = (int)v[p].read;
            if (k < 2) { p++; continue; }                           // (a stack of one is one record: the walk steps by the sizes)
            const int sidx = v[p].sidx;
            const int kept = mc_sum_evalue(T, v, p, p + k, (int)(X.off[sidx + 1] - X.off[sidx]), tmp) - p;
            v[p].read = (uint32_t)k; v[p].chrono = (uint32_t)kept;
            shrank |= kept != k; any = true;
            p += k;
        }
        if (any && shrank) {
            int w = 0;
            for (int p = 0; p < vn;) {
                const int k = (int)v[p].read < 2 ? 1 : (int)v[p].read, kept = (int)v[p].read < 2 ? 1 : (int)v[p].chrono;
                if (w != p) for (int j = 0; j < kept; j++) v[w + j] = v[p + j];
                w += kept; p += k;
            }
            vn = w;
        }
    }
    MC_FR_TICK(0);
    for (int i = 0; i < vn; i++) { items[i].k = v[i].loge; items[i].i = (uint32_t)i; items[i].pad = 0; }
    MC_FR_TICK(1);
    S::sort(items, vn);                              // std::sort by log E (PrintRes), on (key, index) items
    MC_FR_TICK(2);
    int nrows = 0;
    best->read = read_id; best->family = -1; best->aln = 0; best->target_len = 0; best->bits = 0.0;
    while (nrows < vn && nrows < MC_MAX_M8 && v[items[nrows].i].loge < T.loge_thr) nrows++;      // PrintRes: at most 500 rows, log E below the threshold
    // MergeRes re-sorts the printed rows with std::partial_sort (a heap sort) keyed by the PRINTED log E: same heap on the
    // (key, index) items, then the rows are written once, in their final order
    for (int i = 0; i < nrows; i++) items[i].k = mc_round6(v[items[i].i].loge);
    MC_FR_TICK(3);
    S::heap(items, nrows);
    MC_FR_TICK(4);
    (void)krows;
    for (int i = 0; i < nrows; i++) {                // the row is written and classified from the same registers (a thread walks global memory alone: reading it back was a trip per row)
        McRow r;
        mc_fill_row(T, read_id, v[items[i].i], r);
        rows[i] = r;
        int fam = marker_family[r.subject], tl = (int)(X.off[r.subject + 1] - X.off[r.subject]);
        int nmatch = r.frame;
        if (mc_row_passes(P, r, fam, tl, nmatch)) {
            if (best->family < 0 || best->bits < r.bits) { best->family = fam; best->aln = r.alnlen; best->target_len = tl; best->bits = r.bits; }
        }
    }
    MC_FR_TICK(5);
    MC_FR_TICK(6);
    MC_FR_END;
    return nrows;
}
MC_HDN int mc_finish_stacked(const McTables &T, const McIndex &X, const McClassPars &P, const int32_t *marker_family,
                             int read_id, McHsp *v, int vn, McHsp *tmp, McRow *rows, double *krows, McSortItem *items, McBestHit *best)
{
    return mc_finish_stacked_t<McSortsOut>(T, X, P, marker_family, read_id, v, vn, tmp, rows, krows, items, best);
}
// in[0, n): the read's HSPs sorted by (subject, chrono); v: n entries of scratch.  (The test-only emulation's entry; the kernels
// start from the stacks.)
MC_HDN int mc_finish_read(const McTables &T, const McIndex &X, const McClassPars &P, const int32_t *marker_family,
                          int read_id, const McHsp *in, int n, McHsp *v, McHsp *tmp, McRow *rows, double *krows, McSortItem *items, McBestHit *best)
{
    const int vn = mc_build_stacks(in, n, v);
    return mc_finish_stacked(T, X, P, marker_family, read_id, v, vn, tmp, rows, krows, items, best);
}
