// k_gapped.h - stage B: gapped X-drop extension (AlignGapped@0x40a550): one DP per distinct ungapped segment (k_gap_dedupe),
// a lane per flank with the DP rows in LDS (k_gapped_lds), full-size rows as the last resort (k_gapped), the HSPs (k_gap_emit).
#pragma once
#include "mc_hip_common.h"

// Gap tasks are massively redundant: a read that really comes from a marker gene hits every seed of its diagonal, and the
// ungapped X-drop extension of all of them ends in the same segment - same read, frame, subject, start and end.  The gapped
// extension of both flanks depends on nothing else, so it is computed once per distinct segment (2.7 x fewer DPs on reads of
// real genomes) and every task of the group gets its own HSP from the leader's result (the reference keeps them all until
// CalRes compares coordinates; so do we).  Grouping: one open-addressing table of 64-bit entries (tag | task index + 1),
// claimed with a CAS; equal tags are verified on the task records themselves.  Which member of a group becomes its leader
// depends on timing; the results do not.
// The unit of DP work is ONE FLANK of a distinct segment (item = 2 x leader task + side): the two flanks of a task have
// unrelated sizes (a seed near the read's left end has a long right flank), and a wave whose lanes run flank loops of
// different lengths one after the other idles most of the time.  Items are ordered by their number of DP rows.
struct McFlankOut { int16_t gain, c1, c2, ident, steps, runs, gapcols, over; };   // what one flank added (16 B)

// side 0: right flank, walked forwards; side 1: left flank, walked backwards in place.  Returns false when the reference does
// not extend that flank (AlignSeqs 0x413599, 0x4135a9: more than 2 residues must remain on both sequences).
struct McFlank { int qoff, doff, st, n1, n2; };
__device__ __forceinline__ bool mc_flank_of(const McGapTask &g, int qlen, int dlen, int side, McFlank &f)
{
    if (side == 0) {
        const int qend = g.qfwd + g.qp + g.L, dend = g.qfwd + g.dp + g.L;
        f.qoff = qend; f.doff = dend; f.st = 1; f.n1 = qlen - qend; f.n2 = dlen - dend;
    } else {
        const int qleft = g.qp - g.qbwd, dleft = g.dp - g.qbwd;
        f.qoff = qleft - 1; f.doff = dleft - 1; f.st = -1; f.n1 = qleft; f.n2 = dleft;
    }
    return f.n1 > 2 && f.n2 > 2;
}

__device__ __forceinline__ bool mc_gap_same_segment(const McGapTask &a, const McGapTask &b)
{
    return a.read == b.read && a.sidx == b.sidx && (a.chrono >> 25) == (b.chrono >> 25) && a.qp - a.qbwd == b.qp - b.qbwd && a.dp - a.qbwd == b.dp - b.qbwd &&
           a.qp + a.L + a.qfwd == b.qp + b.L + b.qfwd;
}

// consecutive slots of a global counter for n (0..2) entries per thread, one atomic per 256-thread workgroup
__device__ __forceinline__ uint32_t mc_block_alloc_n(uint32_t *counter, uint32_t n)
{
    __shared__ uint32_t wcnt[4], wbase[4];
    const int lane = mc_lane(), wv = threadIdx.x >> 6;
    uint32_t incl = n;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(incl, d); if (lane >= d) incl += y; }
    if (lane == 63) wcnt[wv] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t c0 = wcnt[0], c1 = wcnt[1], c2 = wcnt[2], c3 = wcnt[3], tot = c0 + c1 + c2 + c3;
        const uint32_t b = tot ? atomicAdd(counter, tot) : 0u;
        wbase[0] = b; wbase[1] = b + c0; wbase[2] = b + c0 + c1; wbase[3] = b + c0 + c1 + c2;
    }
    __syncthreads();
    const uint32_t r = wbase[wv] + incl - n;
    __syncthreads();
    return r;
}

// groups the tasks (leader[p] = first task of p's segment to claim the table slot) and lists the flanks of the leaders with
// their sort keys (1 + DP rows, below 1024)
__global__ void __launch_bounds__(256) k_gap_dedupe(McIndex X, int L, const McGapTask *__restrict__ gaps, uint32_t ngaps, unsigned long long *tab, uint32_t mask, uint32_t *leader,
                                                    uint32_t *key, uint32_t *item, uint32_t *counters)
{
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    uint32_t n = 0;
    McFlank fr, fl;
    bool hr = false, hl = false;
    if (p < ngaps && gaps[p].read == MC_TASK_NONE) leader[p] = p;         // (padding of a wave's last block: k_eval_seeds)
    else if (p < ngaps) {
        const McGapTask g = gaps[p];
        unsigned long long h = ((unsigned long long)g.read << 32) ^ ((unsigned long long)g.sidx << 12) ^ (unsigned long long)(g.chrono >> 25);
        h ^= ((unsigned long long)(uint16_t)(g.qp - g.qbwd) << 48) ^ ((unsigned long long)(uint16_t)(g.dp - g.qbwd) << 20) ^ ((unsigned long long)(uint16_t)(g.qp + g.L + g.qfwd) << 3);
        h *= 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
        const unsigned long long mine = (h & ~0x7FFFFFFull) | (unsigned long long)(p + 1);          // tag: the upper 37 bits of the hash
        uint32_t slot = (uint32_t)h & mask, who = p;
        bool lead = false;
        for (;;) {
            unsigned long long e = tab[slot];
            if (e == 0) e = atomicCAS(&tab[slot], 0ull, mine);
            if (e == 0) { lead = true; break; }
            if ((e & ~0x7FFFFFFull) == (mine & ~0x7FFFFFFull)) {
                const uint32_t q = (uint32_t)(e & 0x7FFFFFFull) - 1;
                if (mc_gap_same_segment(g, gaps[q])) { who = q; break; }
            }
            slot = (slot + 1) & mask;
        }
        leader[p] = who;
        if (lead) {
            const int frame = (int)(g.chrono >> 25), qlen = (L - frame % 3) / 3, dlen = (int)(X.off[g.sidx + 1] - X.off[g.sidx]);
            hr = mc_flank_of(g, qlen, dlen, 0, fr); hl = mc_flank_of(g, qlen, dlen, 1, fl);
            n = (hr ? 1u : 0u) + (hl ? 1u : 0u);
        }
    }
    uint32_t o = mc_block_alloc_n(&counters[C_ITEMS], n);
    if (hr) { key[o] = 1u + (uint32_t)fr.n1; item[o] = 2 * p; o++; }
    if (hl) { key[o] = 1u + (uint32_t)fl.n1; item[o] = 2 * p + 1; }
}

// ---- the flanks in descending order of their DP rows (the lanes of a wave of k_gapped_lds then run DPs of similar length) ----------
// A counting sort over the 10-bit keys, three launches (rounds 1 - 3: rocPRIM's radix sort): a histogram (per workgroup in LDS, then
// one global atomic per bin it met), the bins' places from the largest key down, and the scatter - a workgroup counts the keys of
// its own stretch of the list again, reserves its part of every bin with one atomic and deals the places out in LDS.  The order
// inside a bin is left to the atomics: it decides which lane runs which flank, never a result (a flank's result has its own slot).
#define MC_GS_BINS 1024u
__global__ void __launch_bounds__(256) k_gap_sort_hist(const uint32_t *__restrict__ key, const uint32_t *__restrict__ n_p, uint32_t cap, uint32_t *hist)
{
    __shared__ uint32_t h[MC_GS_BINS];
    const uint32_t n = *n_p <= cap ? *n_p : 0u;
    for (uint32_t b = threadIdx.x; b < MC_GS_BINS; b += 256) h[b] = 0;
    __syncthreads();
    const uint32_t per = (n + gridDim.x - 1) / gridDim.x, lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += 256) atomicAdd(&h[key[i] & (MC_GS_BINS - 1)], 1u);
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < MC_GS_BINS; b += 256) if (h[b]) atomicAdd(&hist[b], h[b]);
}
__global__ void __launch_bounds__(1024) k_gap_sort_scan(uint32_t *hist)
{   // hist[b]: the number of keys b -> the place of the first of them, the largest key first
    __shared__ uint32_t s[MC_GS_BINS];
    const uint32_t t = threadIdx.x, v = hist[MC_GS_BINS - 1 - t];
    s[t] = v;
    __syncthreads();
    for (uint32_t d = 1; d < MC_GS_BINS; d <<= 1) { const uint32_t y = t >= d ? s[t - d] : 0u; __syncthreads(); s[t] += y; __syncthreads(); }
    hist[MC_GS_BINS - 1 - t] = s[t] - v;
}
__global__ void __launch_bounds__(256) k_gap_sort_scatter(const uint32_t *__restrict__ key, const uint32_t *__restrict__ item, const uint32_t *__restrict__ n_p, uint32_t cap, uint32_t *base, uint32_t *out)
{
    __shared__ uint32_t h[MC_GS_BINS], at[MC_GS_BINS];
    const uint32_t n = *n_p <= cap ? *n_p : 0u;
    for (uint32_t b = threadIdx.x; b < MC_GS_BINS; b += 256) h[b] = 0;
    __syncthreads();
    const uint32_t per = (n + gridDim.x - 1) / gridDim.x, lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += 256) atomicAdd(&h[key[i] & (MC_GS_BINS - 1)], 1u);
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < MC_GS_BINS; b += 256) { const uint32_t c = h[b]; if (c) at[b] = atomicAdd(&base[b], c); h[b] = 0; }
    __syncthreads();
    for (uint32_t i = lo + threadIdx.x; i < hi; i += 256) { const uint32_t b = key[i] & (MC_GS_BINS - 1); out[at[b] + atomicAdd(&h[b], 1u)] = item[i]; }
}

// every gap task -> its HSP, from the flank results of its group's leader
__global__ void __launch_bounds__(256) k_gap_emit(const McTables *__restrict__ T, McIndex X, int L, const McGapTask *__restrict__ gaps, uint32_t ngaps, const uint32_t *__restrict__ leader,
                                                  const McFlankOut *__restrict__ fout, McHsp *hsps, uint32_t cap_hsps, uint32_t *counters, const McClassPars *__restrict__ P,
                                                  const int32_t *__restrict__ fam, uint8_t *cand, uint64_t *hkeys, uint8_t *low, uint64_t *hplace)
{
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    bool keep = false;
    McHsp h;
    if (p < ngaps && gaps[p].read != MC_TASK_NONE) {                      // (not the padding of a wave's last block)
        const McGapTask g = gaps[p];
        const uint32_t ld = leader[p];
        const int frame = (int)(g.chrono >> 25), qlen = (L - frame % 3) / 3, dlen = (int)(X.off[g.sidx + 1] - X.off[g.sidx]);
        int score = g.score, nmatch = g.nmatch, qfwd = g.qfwd, dfwd = g.qfwd, qbwd = g.qbwd, dbwd = g.qbwd, alnlen = g.qfwd + g.L + g.qbwd, gapopens = 0, gaptotal = 0;
        McFlank f;
        if (mc_flank_of(g, qlen, dlen, 0, f)) {
            const McFlankOut R = fout[2 * (size_t)ld];
            if (R.gain > 0) { score += R.gain; nmatch += R.ident; qfwd += R.c1; dfwd += R.c2; alnlen += R.steps; gapopens += R.runs; gaptotal += R.gapcols; }
        }
        if (mc_flank_of(g, qlen, dlen, 1, f)) {
            const McFlankOut R = fout[2 * (size_t)ld + 1];
            if (R.gain > 0) { score += R.gain; nmatch += R.ident; qbwd += R.c1; dbwd += R.c2; alnlen += R.steps; gapopens += R.runs; gaptotal += R.gapcols; }
        }
        h.read = g.read; h.chrono = g.chrono;
        keep = mc_make_hsp(*T, L, frame, g, qfwd, dfwd, qbwd, dbwd, score, nmatch, alnlen, gapopens, gaptotal, &h);
        if (keep && cand && mc_hsp_can_classify(*T, *P, X, fam, h)) cand[h.read] = 1;
        if (keep && h.loge < T->loge_thr) low[h.read] = 1;
    }
    const uint32_t o = mc_block_alloc(&counters[C_HSPS], keep);
    if (keep) { if (o < cap_hsps) { hsps[o] = h; hkeys[o] = MC_HSP_KEY(h); hplace[o] = MC_HSP_PLACE(h); } else counters[C_OVERFLOW] = 2; }
}

#define MC_GAP_W 1200   // columns of the full-size DP workspace (markers are <= 1183 aa, checked in mc_open)
#define MC_GAP_WIN 36   // columns of the LDS window of the first launch (18 KB per wave: eight waves per CU; per 1 M reads of 150 / 300 bp, first + second launch:
                        // 40 columns x 7 waves 2.00 + 0.41 / 9.49 + 1.05 ms, 36 x 8: 1.73 + 0.44 / 8.19 + 1.64, 32 x 9: 1.85 + 0.81 / 8.43 + 5.89)
#define MC_GAP_WIN2 64  // ... of the second one, for the flanks whose band left the first (32 KB per wave)
#define MC_GAP_LANES2 64 // lanes of a wave that take flanks in the second launch (per 1 M reads of 300 bp behind a 36-column first launch: 16 lanes 1.65 ms, 32: 1.43, 64: 0.93)

__device__ __forceinline__ McFlankOut mc_flank_out(const McGapResult &R)
{
    McFlankOut o;
    o.gain = (int16_t)R.gain; o.c1 = (int16_t)R.c1; o.c2 = (int16_t)R.c2; o.ident = (int16_t)R.ident; o.steps = (int16_t)R.steps; o.runs = (int16_t)R.runs;
    o.gapcols = (int16_t)R.gapcols; o.over = (int16_t)R.overflow;
    return o;
}

// Gapped extension with full-size DP rows in global memory (24 bytes per column, one row set per thread): the last resort for
// the flanks whose band leaves both LDS windows of k_gapped_lds.
__global__ void __launch_bounds__(128) k_gapped(const McTables *__restrict__ T, McIndex X, const uint8_t *__restrict__ frames, int FP, int L,
                                                const McGapTask *__restrict__ gaps, const uint32_t *__restrict__ list, const uint32_t *__restrict__ nitems_p, McFlankOut *fout,
                                                uint32_t *counters, McGapCell *ws, int cap)
{
    __shared__ McHot hot;
    const uint32_t nitems = *nitems_p;                            // (a device-side count: usually 0 - nothing left the windows)
    if (nitems == 0) return;
    mc_load_hot(&hot, T);
    __syncthreads();
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, nthreads = gridDim.x * blockDim.x;
    McGapCell *C = ws + (size_t)tid * cap;
    for (uint32_t k0 = tid; k0 < nitems; k0 += nthreads) {
        const uint32_t it = list[k0];
        const McGapTask g = gaps[it >> 1];
        const int frame = (int)(g.chrono >> 25), qlen = (L - frame % 3) / 3;
        const uint32_t o0 = X.off[g.sidx];
        McFlank f;
        (void)mc_flank_of(g, qlen, (int)(X.off[g.sidx + 1] - o0), (int)(it & 1), f);
        const McGapResult R = mc_align_gapped(hot, frames + ((int64_t)g.read * 6 + frame) * FP + f.qoff, f.st, X.res + o0 + f.doff, f.st, f.n1, f.n2, C, cap);
        if (R.overflow) counters[C_OVERFLOW] = 5;
        fout[it] = mc_flank_out(R);
    }
}

// Gapped extension: one lane per flank item with its DP rows in LDS.  The extension (mc_gap_begin / mc_gap_row, mc_core.h) keeps
// only the live band - a circular window of W columns, 12 bytes per column: the two scores (16 + 16 bits) in one word, the two
// path-statistics words, the subject residue in the spare byte of the second - nothing of the DP touches global memory.
// Layout: word (slot, lane) of a wave's window sits at slot * 64 + lane, so whatever slots the 64 lanes are working on they fall
// into 64 different banks.  A flank whose band is wider than the window (0.3 % of the flanks of 150 bp reads at W = 36) goes to
// the retry list: the same kernel with a 64-column window, and behind that k_gapped with full-size rows in global memory.
//
// PERSISTENT LANES.  How long a flank takes is not known before it ends: its DP rows (the sort key) are only an upper bound -
// the X-drop rule ends most extensions early - so 64 flanks of equal key dealt to the 64 lanes of a wave keep 58 % (150 bp) /
// 39 % (300 bp) of the lanes busy even with perfectly balanced rows (measured on the host: cells per flank, tests/emul).  So a
// lane does not wait for its wave: the wave loops over DP ROWS, and whenever MC_GAP_REFILL lanes have ended their flanks they
// start their next ones together (row 0 is set up by all of them at once).  Wave w of G owns items w, w + G, w + 2 G ... of the
// list, which is in descending order of DP rows: every wave sees the same mix, longest first.
// FETCHED AHEAD.  A flank starts with a chain of dependent global reads - item id, task record, subject offsets, the subject
// residues of row 0 - each a memory round trip that the whole wave would wait for.  So a lane claims its next item the moment it
// starts one, and walks that chain one link per loop iteration (a DP row of the others) while it works: when its flank ends the
// next one is ready in registers.  The same inside a row: the query residue of the next row and the subject residues the right
// growth will need are requested at the row's start (mc_gap_row).
#define MC_GAP_REFILL 8
// A DP column in 8 bytes (mc_gap_pack / mc_gap_unpack, mc_core.h): 20 KB per wave at 40 columns - seven waves per CU instead of the five
// that 12-byte columns allowed, and the kernel's speed is proportional to the waves a CU holds (it waits on its own chains of
// dependent instructions: 2 / 3 / 4 / 5 waves per CU ran 5.6 / 3.8 / 3.1 / 2.45 ms).  One 64-bit LDS access per cell and direction.
template <int W>
struct McGapLds {
    uint2 *cell;                                                   // this lane's column 0; column c at cell[c * 64]
    uint32_t ovf;                                                  // a path statistic left its packed field (nothing the kernel cannot redo wider)
    __device__ __forceinline__ void load(int c, int &H, int &D, uint32_t &PH, uint32_t &PD) const
    {
        const uint2 w = cell[c * 64];
        mc_gap_unpack(w.x, w.y, H, D, PH, PD);
    }
    __device__ __forceinline__ void store(int c, int H, int D, uint32_t PH, uint32_t PD)
    {
        uint2 w;
        ovf |= mc_gap_pack(H, D, PH, PD, w.x, w.y);
        cell[c * 64] = w;
    }
    __device__ __forceinline__ int loadH(int c) const { return (int)(cell[c * 64].x << 20) >> 20; }
};

template <int W, int LANES>
__global__ void __launch_bounds__(64) k_gapped_lds(const McTables *__restrict__ T, McIndex X, const uint8_t *__restrict__ frames, int FP, int L,
                                                   const McGapTask *__restrict__ gaps, const uint32_t *__restrict__ list, const uint32_t *__restrict__ nitems_p, McFlankOut *fout,
                                                   uint32_t *retry_count, uint32_t *retry, int refill, uint32_t *take)
{
    __shared__ McHot hot;
    __shared__ uint2 win[W * 64];
    const uint32_t nitems = *nitems_p;                            // device-side count
    if (nitems == 0) return;
    mc_load_hot(&hot, T);
    __syncthreads();
    const int lane = threadIdx.x;
    McGapLds<W> ws; ws.cell = win + lane; ws.ovf = 0;
    // LANES < 64 (the retry launch: few, large flanks - its run time is that of the longest chain of them in one lane): only the
    // first LANES lanes of a wave take items, so that the items spread over all the waves the GPU holds
    const bool mine = lane < LANES;
    const uint32_t G = gridDim.x, w0 = blockIdx.x;
    // The items of this wave: windows of WS consecutive items of the list (which is in descending order of DP rows), the first one by the
    // wave's number, every further one from a counter, asked for when half of the window before is used up - longest first, to whoever
    // is free.  (Dealt out in turn - wave w took items w, w + G, ... - the waves were resident for 85 % of the launch: SQ_WAVE_CYCLES.)
    constexpr uint32_t WS = (uint32_t)LANES;                       // (the retry launch: as many items as lanes that take them - its few items spread over all the waves)
    uint32_t wpos = w0 * WS < nitems ? w0 * WS : nitems, wend = wpos + WS < nitems ? wpos + WS : nitems;
    uint32_t pend = 0;                                             // lane 0: the number of the next window
    bool asked = false, dry = false;                               // a window has been asked for / there are no more
    const unsigned long long lt = (1ull << lane) - 1;
    const int REFILL = LANES < refill ? 1 : refill;
    // the flank being extended
    bool active = false;
    uint32_t it = 0;
    McGapState S;
    // the flank fetched ahead.  nstage: 0 nothing claimed, 1 item id on its way, 2 task record, 3 subject offsets, 4 row-0 residues, 5 ready
    int nstage = 0;
    uint32_t nit = 0, no0 = 0, no1 = 0, nraw[sizeof(McGapTask) / 4];
    McFlank nf;
    const uint8_t *ns1 = nullptr, *ns2 = nullptr;
    uint64_t nlo = 0, nhi = 0;
    uint32_t nx0 = 0;
    nf.qoff = nf.doff = nf.st = nf.n1 = nf.n2 = 0;
    for (;;) {
        // Everything this wave requested from global memory during the last iteration - a DP row ago - has arrived by now: said once,
        // here, so that no later use waits for it together with the requests of THIS iteration (the counter is in order).
        __builtin_amdgcn_s_waitcnt(0x0F70);                        // vmcnt(0)
        bool fin = false;
        {   // ---- the idle lanes whose next flank is ready start it - MC_GAP_REFILL of them together, or when nothing else is left to do
            const bool ready = mine && !active && nstage == 5;
            const unsigned long long rm = __ballot(ready);
            if (rm && (__popcll(rm) >= REFILL || (dry && wpos >= wend) || __ballot(active) == 0)) {
                if (ready) {
                    it = nit; active = true; nstage = 0; ws.ovf = 0;
                    fin = !mc_gap_begin(hot, S, ns1, ns2, nf.st, nf.n1, nf.n2, ws, W, true, nlo, nhi, (int)nx0);
                }
            }
        }
        // ---- the flank behind it: one link of the chain per iteration (what the link needs was requested an iteration ago)
        if (nstage == 4) {                                          // the 16 residues in walking order, one byte each
            if (nf.st < 0) { const uint64_t a = nlo; nlo = __builtin_bswap64(nhi); nhi = __builtin_bswap64(a); }
            nstage = 5;
        } else if (nstage == 3) {
            McGapTask ng;
            __builtin_memcpy(&ng, nraw, sizeof ng);
            const int frame = (int)(ng.chrono >> 25), qlen = (L - frame % 3) / 3;
            (void)mc_flank_of(ng, qlen, (int)(no1 - no0), (int)(nit & 1), nf);
            ns1 = frames + ((int64_t)ng.read * 6 + frame) * FP + nf.qoff; ns2 = X.res + no0 + nf.doff;
            const uint8_t *lowest = nf.st > 0 ? ns2 : ns2 - 15;    // 16 bytes in memory order (the residue array is padded by 64 bytes at both ends)
            __builtin_memcpy(&nlo, lowest, 8); __builtin_memcpy(&nhi, lowest + 8, 8);
            nx0 = ns1[0];
            nstage = 4;
        } else if (nstage == 2) {
            const uint32_t sidx = nraw[offsetof(McGapTask, sidx) / 4];
            no0 = X.off[sidx]; no1 = X.off[sidx + 1];
            nstage = 3;
        } else if (nstage == 1) {
            const uint32_t *gp = (const uint32_t *)(gaps + (nit >> 1));
#pragma unroll
            for (int k = 0; k < (int)(sizeof(McGapTask) / 4); k++) nraw[k] = gp[k];
            nstage = 2;
        }
        {   // claim: the lanes without a next item take the next ones of the wave's share
            const bool want = mine && nstage == 0;
            const unsigned long long cm = __ballot(want);
            if (cm && wpos >= wend && !dry) {                       // the window is used up: on to the next one
                if (!asked && lane == 0) pend = atomicAdd(take, 1u);
                const uint32_t base = (G + (uint32_t)__builtin_amdgcn_readfirstlane((int)pend)) * WS;
                asked = false;
                if (base >= nitems) { dry = true; wpos = wend = nitems; }
                else { wpos = base; wend = base + WS < nitems ? base + WS : nitems; }
            }
            if (cm && wpos < wend) {
                const uint32_t k = wpos + (uint32_t)__popcll(cm & lt);
                if (want && k < wend) { nit = list[k]; nstage = 1; }
                wpos += (uint32_t)__popcll(cm);
                if (wpos > wend) wpos = wend;
            }
            if (!asked && !dry && wend - wpos <= WS / 2) { if (lane == 0) pend = atomicAdd(take, 1u); asked = true; }
        }
        // ---- one DP row of every flank in progress
        if (active && !fin) fin = mc_gap_row(hot, S, ws, W);
        if (fin && ws.ovf) S.over = 1;                              // (more than 31 gap runs on a live path: redone with the wider launch, in the end with full-size cells)
        if (fin) { fout[it] = mc_flank_out(mc_gap_result(S)); active = false; }
        const bool over = fin && S.over != 0;
        const uint32_t ro = mc_wave_alloc(retry_count, over);      // band left the window: the flank is redone with a wider one
        if (over) retry[ro] = it;
        if (dry && wpos >= wend && __ballot(active || nstage != 0) == 0) break;
    }
}
