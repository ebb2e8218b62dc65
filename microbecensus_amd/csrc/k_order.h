// k_order.h - stage C: the HSPs of every read together, ordered by (subject, hit order), and CalRes' stacks of the reads that
// can print anything (k_bin_*, mc_scan_u32, k_order_*) - what the reference's multimap<(query, subject)> does (CalRes 0x407c70).
#pragma once
#include "mc_hip_common.h"

// ---- HSPs into per-read segments, ordered by (subject, hit order) ----------------------------------------------------------------
// The reference keeps a read's HSPs in a multimap keyed by (query, subject) (`CalRes` insert 0x407c70, `PrintRes@0x409310` walks it
// subject by subject): the finishing kernels need every read's HSPs together, ordered by subject and - inside a subject - by the
// order in which the reference would have found them (chrono).  Rounds 1 - 3 got there with a 64-bit radix sort of ALL HSPs
// (rocPRIM, 8 passes over 45 M keys per 2 M reads).  But the producers emit the HSPs of a read close together (a read's seed hits
// are consecutive in the task pool), nine reads in ten print nothing whatever the order of their HSPs, and a read has 23 HSPs on
// average.  So: (1) count the HSPs per read and scan the counts (k_bin_count, mc_scan_*), (2) move every HSP's key and pool slot -
// 12 bytes, not the 48-byte record - to its read's segment (k_bin_scatter; both with ONE atomic per run of consecutive HSPs of the
// same read in the pool), (3) order each segment by
// (subject, hit order) - every HSP's rank inside its segment is the number of smaller keys there, counted in LDS - and decide
// whether the read can print anything: a workgroup per 64 reads for the segments of up to 64 HSPs (k_order_light), a wave per read
// for the longer ones (k_order_heavy); (4) only the records of the reads that can print are fetched from the pool, in order (k_order_copy).
// A read is MARKED (nrow_of = 1: the finishing kernels take it) when one of its HSPs has log E below the threshold (low[read], set
// by the kernel that made the HSP) or two DIFFERENT HSPs lie on one subject (sum statistics may lower the group's E; HSPs of a
// subject with the same frame and coordinates are one HSP found from several seeds: CalRes keeps the best of them, printed only if
// its own log E is below the threshold).  Marking more reads than that is harmless (a marked read that prints nothing finishes with
// 0 rows), only slower.
// hkeys[slot] = read << 43 | subject << 28 | hit order, written beside every HSP by the kernel that makes it (~0: padding).
#ifndef MC_BIN_LIGHT
#define MC_BIN_LIGHT 32                    // segments up to this long are ordered by k_order_light, longer ones by k_order_heavy
#endif
__device__ __forceinline__ void mc_bin_runs(bool valid, uint32_t read, int lane, bool &head, int &hl, uint32_t &len)
{   // consecutive lanes of the wave with the same read form a run: head = its first lane, hl = the head's lane, len = its length (valid lanes only)
    const uint32_t pr = (uint32_t)__shfl_up((int)read, 1);
    const bool pv = (bool)__shfl_up((int)valid, 1);
    head = valid && (lane == 0 || !pv || pr != read);
    const unsigned long long hm = __ballot(head), sm = __ballot(!valid || head);
    const unsigned long long below = hm & ((2ull << lane) - 1ull);
    hl = below ? 63 - __builtin_clzll(below) : 0;
    const unsigned long long above = lane < 63 ? (sm & ~((2ull << lane) - 1ull)) : 0ull;
    len = (uint32_t)((above ? __builtin_ctzll(above) : 64) - lane);
}
__global__ void __launch_bounds__(256) k_bin_count(const uint64_t *__restrict__ hkeys, const uint32_t *__restrict__ counters_in, uint32_t cap_hsps, const uint8_t *__restrict__ cand, uint32_t *cnt)
{
    const uint32_t n = counters_in[C_HSPS] <= cap_hsps ? counters_in[C_HSPS] : 0u;
    const int lane = mc_lane();
    for (uint32_t b0 = blockIdx.x * 256u; b0 < n; b0 += gridDim.x * 256u) {
        const uint32_t tid = b0 + threadIdx.x;
        uint64_t key = ~0ull;
        if (tid < n) key = hkeys[tid];
        const uint32_t read = (uint32_t)(key >> 43);
        const bool valid = key != ~0ull && (!cand || cand[read] != 0);          // (~0: padding of a wave's last block)
        bool head; int hl; uint32_t len;
        mc_bin_runs(valid, read, lane, head, hl, len);
        if (head) atomicAdd(&cnt[read], len);
    }
}
__global__ void __launch_bounds__(256) k_bin_scatter(const uint64_t *__restrict__ hkeys, const uint32_t *__restrict__ counters_in, uint32_t cap_hsps,
                                                     const uint8_t *__restrict__ cand, uint32_t *cur, const uint64_t *__restrict__ hplace, uint64_t *keys, uint64_t *places, uint32_t *slots)
{   // cur[read]: where the read's next HSP goes (in: the exclusive scan of the counts; out: the END of every read's segment = the start of the next read's).
    // Only the key, the place word and the pool slot of an HSP move (20 bytes): the 48-byte records stay in the pool until k_order_copy fetches those of the marked reads.
    const uint32_t n = counters_in[C_HSPS] <= cap_hsps ? counters_in[C_HSPS] : 0u;
    const int lane = mc_lane();
    for (uint32_t b0 = blockIdx.x * 256u; b0 < n; b0 += gridDim.x * 256u) {
        const uint32_t tid = b0 + threadIdx.x;
        uint64_t key = ~0ull;
        if (tid < n) key = hkeys[tid];
        const uint32_t read = (uint32_t)(key >> 43);
        const bool valid = key != ~0ull && (!cand || cand[read] != 0);
        bool head; int hl; uint32_t len;
        mc_bin_runs(valid, read, lane, head, hl, len);
        uint32_t base = 0;
        if (head) base = atomicAdd(&cur[read], len);
        base = (uint32_t)__shfl((int)base, hl);
        if (valid) { const uint32_t dst = base + (uint32_t)(lane - hl); keys[dst] = key; places[dst] = hplace[tid]; slots[dst] = tid; }
    }
}
// exclusive scan of n 32-bit counts (n <= 2 M + 1): partial sums of blocks of 1024, the scan of those by one workgroup, the blocks again
#define MC_SCAN_BLK 1024u
__global__ void __launch_bounds__(256) k_scan_sums(const uint32_t *__restrict__ in, uint32_t n, uint32_t *sums)
{
    __shared__ uint32_t w[4];
    const uint32_t i0 = blockIdx.x * MC_SCAN_BLK + threadIdx.x * 4u;
    uint32_t v = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) if (i0 + k < n) v += in[i0 + k];
    for (int d = 32; d > 0; d >>= 1) v += (uint32_t)__shfl_down((int)v, d);
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = w[0] + w[1] + w[2] + w[3];
}
__global__ void __launch_bounds__(1024) k_scan_top(uint32_t *sums, uint32_t nb)
{   // one workgroup: exclusive scan of up to 4096 block sums in place (4 per thread)
    __shared__ uint32_t w[16];
    const int lane = mc_lane(), wv = threadIdx.x >> 6;
    uint32_t x[4], t = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) { const uint32_t i = threadIdx.x * 4u + k; x[k] = i < nb ? sums[i] : 0u; t += x[k]; }
    const uint32_t inc = mc_wave_scan_add(t);
    if (lane == 63) w[wv] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int k = 0; k < wv; k++) base += w[k];
    uint32_t run = base + inc - t;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) { const uint32_t i = threadIdx.x * 4u + k; if (i < nb) sums[i] = run; run += x[k]; }
}
__global__ void __launch_bounds__(256) k_scan_apply(const uint32_t *__restrict__ in, uint32_t n, const uint32_t *__restrict__ sums, uint32_t *out)
{
    __shared__ uint32_t w[4];
    const int lane = mc_lane(), wv = threadIdx.x >> 6;
    const uint32_t i0 = blockIdx.x * MC_SCAN_BLK + threadIdx.x * 4u;
    uint32_t x[4], t = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) { x[k] = i0 + k < n ? in[i0 + k] : 0u; t += x[k]; }
    const uint32_t inc = mc_wave_scan_add(t);
    if (lane == 63) w[wv] = inc;
    __syncthreads();
    uint32_t run = sums[blockIdx.x] + inc - t;
    for (int k = 0; k < wv; k++) run += w[k];
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) { if (i0 + k < n) out[i0 + k] = run; run += x[k]; }
}
// out[i] = sum of in[0 .. i) for i < n (in and out may be the same array); sums: ceil(n / 1024) + 1 words of scratch
static int mc_scan_u32(const uint32_t *in, uint32_t n, uint32_t *out, uint32_t *sums, hipStream_t st)
{
    if (!n) return 0;
    const uint32_t nb = (n + MC_SCAN_BLK - 1) / MC_SCAN_BLK;
    if (nb > 4096) { g_err = "scan of more than 4 M counts"; return -1; }
    k_scan_sums<<<dim3(nb), dim3(256), 0, st>>>(in, n, sums);
    k_scan_top<<<dim3(1), dim3(1024), 0, st>>>(sums, nb);
    k_scan_apply<<<dim3(nb), dim3(256), 0, st>>>(in, n, sums, out);
    return 0;
}

__device__ __forceinline__ bool mc_hsp_same_place(const McHsp *a, const McHsp *b)
{   // frame and the four coordinates: the HSP was found again from another seed (CalRes 0x4082b0-0x408446 keeps one of them)
    return a->frame == b->frame && a->qaas == b->qaas && a->ds == b->ds && a->qaae == b->qaae && a->de == b->de;
}
__device__ __forceinline__ void mc_hsp_copy(McHsp *dst, const McHsp *src)
{
    const uint4 *s = (const uint4 *)src; uint4 *d = (uint4 *)dst;
    const uint4 x0 = s[0], x1 = s[1], x2 = s[2];
    d[0] = x0; d[1] = x1; d[2] = x2;
}
// Light reads (segments of up to MC_BIN_LIGHT HSPs): a workgroup takes 64 consecutive reads - one contiguous stretch of the binned keys -
// and stages the subjects and the (subject << 28 | hit order) keys of their HSPs in LDS.  Nine reads in ten have no HSP below the
// threshold: for their HSPs only the question "is there another HSP on my subject, and is it a different one" is asked (a loop over
// the segment's subjects in LDS; frame and coordinates are compared in global memory, rarely).  The HSPs of the marked reads are
// then ranked inside their segment by counting the smaller keys and copied to their ranks.  The reads with longer segments are
// listed for k_order_heavy.
#define MC_OL_READS 64
#ifndef MC_ORDER_SMALL                     // (a test builds the library with small arrays so that ordinary reads take the paths of the longest ones)
#define MC_ORDER_SMALL 512                 // segments up to this long: a wave per read (two buffers of 4 KB in LDS) ...
#define MC_ORDER_MID 2048                  // ... up to this long (3 reads in 1,000): a workgroup of four waves (two buffers of 16 KB) ...
#define MC_ORDER_LDS 8192                  // ... the few longer ones (0.6 in 1,000): a workgroup of sixteen waves (two buffers of 64 KB; beyond that: blocks of 8192, merged in global memory)
#endif
// the reads whose segments are longer than MC_BIN_LIGHT, listed for k_order_heavy (a thread per read)
__global__ void __launch_bounds__(256) k_order_lists(const uint32_t *__restrict__ heads, uint32_t nreads, uint32_t *counters, uint32_t *heavy, uint32_t *heavy2, uint32_t *heavy3)
{
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    const uint32_t n = r < nreads ? heads[r + 1] - heads[r] : 0u;
    const bool c1 = n > MC_BIN_LIGHT && n <= MC_ORDER_SMALL, c2 = n > MC_ORDER_SMALL && n <= MC_ORDER_MID, c3 = n > MC_ORDER_MID;
    const int idx[3] = {C_ORDER, C_ORDER2, C_ORDER3};
    uint32_t off[3];
    mc_block_alloc_multi<3>(counters, idx, (c1 ? 1u : 0u) | (c2 ? 2u : 0u) | (c3 ? 4u : 0u), off);
    if (c1) heavy[off[0]] = r;
    if (c2) heavy2[off[1]] = r;
    if (c3) heavy3[off[2]] = r;
}
#define MC_KEY43 ((1ull << 43) - 1)
__global__ void __launch_bounds__(256) k_order_light(const uint64_t *__restrict__ keys, const uint64_t *__restrict__ places, const uint32_t *__restrict__ slots, const uint32_t *__restrict__ heads, uint32_t nreads,
                                                    const uint8_t *__restrict__ low, uint32_t *order, uint32_t *gsz, uint32_t *nv, uint32_t *nrow_of)
{
    __shared__ uint64_t key[MC_OL_READS * MC_BIN_LIGHT], plc[MC_OL_READS * MC_BIN_LIGHT];
    __shared__ uint16_t sid[MC_OL_READS * MC_BIN_LIGHT];
    __shared__ uint8_t qof[MC_OL_READS * MC_BIN_LIGHT];            // the read (0 .. 63) of an LDS slot
    __shared__ uint32_t lpos[MC_OL_READS + 1], lhead[MC_OL_READS + 1], lcnt[MC_OL_READS];
    __shared__ uint8_t lmark[MC_OL_READS];
    __shared__ uint32_t gmask[MC_OL_READS], rmask[MC_OL_READS];   // per read: which of its (ordered) HSPs is the first of its subject / of its run
    static_assert(MC_BIN_LIGHT <= 32, "a segment's HSPs are bits of a word");
    const int lane = mc_lane(), wv = threadIdx.x >> 6;
    const uint32_t r0 = blockIdx.x * MC_OL_READS;
    if (wv == 0) {
        const uint32_t r = r0 + (uint32_t)lane;
        uint32_t a = 0, n = 0;
        if (r < nreads) { a = heads[r]; n = heads[r + 1] - a; }
        else a = heads[nreads];
        const bool light = n > 0 && n <= MC_BIN_LIGHT;
        const uint32_t m = light ? n : 0u, inc = mc_wave_scan_add(m);
        lpos[lane] = inc - m; lhead[lane] = a; lcnt[lane] = m;
        lmark[lane] = (light && low[r] != 0) ? 1 : 0;
        gmask[lane] = 0; rmask[lane] = 0;
        if (lane == 63) { lpos[64] = inc; lhead[64] = a + n; }
    }
    __syncthreads();
    const uint32_t T = lpos[64];
    if (T == 0) return;
    const uint32_t A = lhead[0], B = lhead[64];
    for (uint32_t p = A + threadIdx.x; p < B; p += 256) {          // subjects and keys into LDS: one coalesced pass over the stretch's keys
        const uint64_t k = keys[p];
        const uint32_t q = (uint32_t)(k >> 43) - r0;
        if (lcnt[q]) { const uint32_t at = lpos[q] + (p - lhead[q]); qof[at] = (uint8_t)q; sid[at] = (uint16_t)((k >> 28) & 0x7FFFu); key[at] = k & MC_KEY43; plc[at] = places[p]; }
    }
    __syncthreads();
    for (uint32_t at = threadIdx.x; at < T; at += 256) {           // reads without an HSP below the threshold: two different HSPs on one subject?
        const uint32_t q = qof[at];
        if (lmark[q]) continue;
        const uint32_t n = lcnt[q], base = lpos[q], me = at - base;
        const uint32_t s = sid[at];
        uint32_t same = 0;
        for (uint32_t j = 0; j < n; j++) same += (sid[base + j] == s) ? 1u : 0u;
        if (same > 1) {
            const uint64_t mine = plc[at];
            for (uint32_t j = 0; j < n; j++)
                if (sid[base + j] == s && MC_PLACE_OF(plc[base + j]) != MC_PLACE_OF(mine)) { lmark[q] = 1; break; }
        }
        (void)me;
    }
    __syncthreads();
    // the marked reads: every HSP's rank in its segment (kept in registers), then keys, place words and positions in order in LDS
    constexpr int PER = MC_OL_READS * MC_BIN_LIGHT / 256;
    uint64_t rk[PER], rp[PER];
    uint32_t rto[PER];
#pragma unroll
    for (int it = 0; it < PER; it++) {
        const uint32_t at = threadIdx.x + 256u * (uint32_t)it;
        rto[it] = ~0u;
        if (at < T && lmark[qof[at]]) {
            const uint32_t q = qof[at], n = lcnt[q], base = lpos[q], me = at - base;
            const uint64_t k = key[at];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < n; j++) { const uint64_t kj = key[base + j]; rank += (kj < k || (kj == k && j < me)) ? 1u : 0u; }
            rk[it] = k; rp[it] = plc[at]; rto[it] = ((base + rank) << 8) | me;
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < PER; it++) if (rto[it] != ~0u) { const uint32_t to = rto[it] >> 8; key[to] = rk[it]; plc[to] = rp[it]; sid[to] = (uint16_t)(rto[it] & 0xFFu); }   // (sid: now the HSP's position in its binned segment)
    __syncthreads();
    // ... and CalRes' stacks of the marked reads (mc_build_stacks, mc_finish.h): of the consecutive HSPs of one place - a RUN - the best
    // one, a subject's runs newest first, the stack's size with its first record.  A thread per HSP (a thread per read walking its
    // HSPs one after the other was 0.39 of the kernel's 1.02 ms per 2 M reads, and the other ordering kernels ran 0.3 ms longer beside
    // it): every HSP marks in two words of its read whether it starts a subject and whether it starts a run - a segment has at most 32
    // HSPs -, and the first HSP of every run reads everything it needs off those words with a few bit counts.
    for (uint32_t at = threadIdx.x; at < T; at += 256) {
        const uint32_t q = qof[at];
        if (!lmark[q]) continue;
        const uint32_t j = at - lpos[q];
        const bool gh = j == 0 || (key[at] >> 28) != (key[at - 1] >> 28), rh = gh || MC_PLACE_OF(plc[at]) != MC_PLACE_OF(plc[at - 1]);
        if (gh) atomicOr(&gmask[q], 1u << j);
        if (rh) atomicOr(&rmask[q], 1u << j);
    }
    __syncthreads();
    for (uint32_t at = threadIdx.x; at < T; at += 256) {
        const uint32_t q = qof[at];
        if (!lmark[q]) continue;
        const uint32_t base = lpos[q], j = at - base, R = rmask[q];
        if (!((R >> j) & 1u)) continue;                             // (not the first HSP of a run)
        const uint32_t G = gmask[q], n = lcnt[q], a = lhead[q];
        const uint32_t upto = (2u << j) - 1u;                      // bits 0 .. j
        const uint32_t gs = 31u - (uint32_t)__builtin_clz(G & upto);             // the subject's first HSP (bit 0 is always set)
        const uint32_t Ggt = G & ~upto, Rgt = R & ~upto;
        const uint32_t ge = Ggt ? (uint32_t)__builtin_ctz(Ggt) : n, j2 = Rgt ? (uint32_t)__builtin_ctz(Rgt) : n;   // the next subject's / the next run's first HSP
        const uint32_t lt_gs = (1u << gs) - 1u, lt_j = (1u << j) - 1u, lt_ge = ge >= 32u ? 0xFFFFFFFFu : (1u << ge) - 1u;
        const uint32_t out = (uint32_t)__builtin_popcount(R & lt_gs), run = (uint32_t)__builtin_popcount(R & lt_j & ~lt_gs), kg = (uint32_t)__builtin_popcount(R & lt_ge & ~lt_gs);
        uint32_t bestj = j;
        for (uint32_t t = j + 1; t < j2; t++) if (MC_SCORE_OF(plc[base + t]) > MC_SCORE_OF(plc[base + bestj])) bestj = t;
        const uint32_t o = a + out + kg - 1 - run;
        order[o] = slots[a + sid[base + bestj]];
        gsz[o] = run == kg - 1 ? kg : 0u;
        if (j == 0) { nv[r0 + q] = (uint32_t)__builtin_popcount(R); nrow_of[r0 + q] = 1u; }
    }
}
// v[i] = the HSP that belongs at place i of the stacks (order[i]: its pool slot; ~0: nothing - the place of an unmarked read's HSP
// or of a duplicate), with the size of its subject's stack in .read (0 behind the stack's first record)
__global__ void __launch_bounds__(256) k_order_copy(const uint32_t *__restrict__ order, const uint32_t *__restrict__ gsz, const McHsp *__restrict__ hsps, const uint32_t *__restrict__ heads, uint32_t nreads, McHsp *v)
{
    const uint32_t total = heads[nreads];
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const uint32_t sl = order[i];
        if (sl != ~0u) {
            const uint4 *s4 = (const uint4 *)(hsps + sl); uint4 *d4 = (uint4 *)(v + i);
            uint4 x0 = s4[0];
            const uint4 x1 = s4[1], x2 = s4[2];
            x0.x = gsz[i];                                          // (.read)
            d4[0] = x0; d4[1] = x1; d4[2] = x2;
        }
    }
}
// A wave per read with more HSPs (reads of marker genes: hundreds of HSPs on homologous markers), a workgroup of eight waves for the
// few with more than 512 (4 reads in 1,000, with a quarter of all HSPs): merge sort of the items (subject << 28 | hit order) << 21 |
// position in LDS (up to MC_ORDER_LDS; longer segments in global scratch); marked like the light reads.
template <int NT> __device__ __forceinline__ void mc_group_sync() { if (NT == 64) mc_wave_sync(); else __syncthreads(); }
__device__ __forceinline__ uint64_t mc_wave_sort64(uint64_t v, int lane)
{   // bitonic sort of one item per lane, ascending by lane, in registers
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1)
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const uint64_t o = __shfl_xor(v, j);
            const bool keep_min = ((lane & k) == 0) == ((lane & j) == 0);
            v = keep_min ? (v < o ? v : o) : (v < o ? o : v);
        }
    return v;
}
// Merge sort of m items (a power of two >= 64, all different) by NT threads: chunks of 64 in registers, then log2(m / 64) passes in
// which every item finds its place in the merged run by a binary search in the partner run - a pass is one barrier, where the
// bitonic network has log2(m) (log2(m) + 1) / 2 of them (78 for the 4096 items of a read of a marker gene with 2,700 HSPs).  A
// thread searches for four items at a time: the four chains of dependent reads run side by side.
// One pass: runs of w items of x (sorted) -> runs of 2 w items of y.
template <int NT, class PTR>
__device__ __forceinline__ void mc_merge_pass(PTR x, PTR y, uint32_t m, uint32_t w, int tid)
{
    for (uint32_t i0 = (uint32_t)tid; i0 < m; i0 += 4 * NT) {
        uint64_t v[4];
        uint32_t lo[4], hi[4], pb[4], at[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const uint32_t i = i0 + (uint32_t)c * NT;
            const bool ok = i < m;
            v[c] = ok ? x[i] : 0ull;
            const uint32_t run = i / w;
            pb[c] = (run ^ 1u) * w; at[c] = (run >> 1) * 2 * w + (i & (w - 1));
            lo[c] = 0; hi[c] = ok ? w : 0u;                            // the number of items of the partner run below v
        }
        for (uint32_t span = w; span > 0; span >>= 1) {                 // (a range of w + 1 answers: log2(w) + 1 halvings)
#pragma unroll
            for (int c = 0; c < 4; c++)
                if (lo[c] < hi[c]) { const uint32_t mid = (lo[c] + hi[c]) >> 1; if (x[pb[c] + mid] < v[c]) lo[c] = mid + 1; else hi[c] = mid; }
        }
#pragma unroll
        for (int c = 0; c < 4; c++) if (i0 + (uint32_t)c * NT < m) y[at[c] + lo[c]] = v[c];
    }
}
// x holds the items, y is a second buffer of the same size; returns the buffer that holds the result.
template <int NT, class PTR>
__device__ __forceinline__ PTR mc_group_mergesort(PTR x, PTR y, uint32_t m, int tid)
{
    const int lane = tid & 63;
    for (uint32_t c = (uint32_t)(tid >> 6) * 64u; c < m; c += NT) x[c + lane] = mc_wave_sort64(x[c + lane], lane);
    mc_group_sync<NT>();
    for (uint32_t w = 64; w < m; w <<= 1) {
        mc_merge_pass<NT>(x, y, m, w, tid);
        mc_group_sync<NT>();
        PTR t = x; x = y; y = t;
    }
    return x;
}
#define MC_ITEM_OF(keys, k, n) ((k) < (n) ? (((keys)[k] & MC_KEY43) << 21) | (uint64_t)(k) : (~0ull << 21) | (uint64_t)(k))   // (padding: behind every HSP, all different)
// From a read's sorted items x (y: the other buffer, R: n counters): is the read marked, and if so CalRes' stacks (mc_build_stacks,
// mc_finish.h) as pool slots in order[0, runs) with the stack sizes in gsz - a run = consecutive HSPs of one subject with the same
// place (the best of them stays), a subject's runs newest first.  R[k] = number of runs that start at or in front of item k.
template <int NT, class PTR, class RPTR>
__device__ __forceinline__ void mc_order_heavy_out(PTR x, PTR y, RPTR R, const uint32_t *__restrict__ slots, const uint64_t *__restrict__ places, uint32_t n, bool marked,
                                                   uint32_t *__restrict__ order, uint32_t *__restrict__ gsz, uint32_t *nrow, uint32_t *nv, int tid, uint32_t *s_w)
{
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll 4
    for (uint32_t k = (uint32_t)tid; k < n; k += NT) y[k] = places[(uint32_t)(x[k] & 0x1FFFFFu)];
    if (NT > 64 && tid == 0) s_w[16] = 0;
    mc_group_sync<NT>();
    if (!marked) {                                                 // no HSP below the threshold: two different HSPs on one subject? (neighbours now)
        bool diff = false;
        for (uint32_t k = (uint32_t)tid + 1; k < n && !diff; k += NT) diff = (x[k - 1] >> 49) == (x[k] >> 49) && MC_PLACE_OF(y[k - 1]) != MC_PLACE_OF(y[k]);
        if (NT == 64) marked = __ballot(diff) != 0;
        else { if (diff) s_w[16] = 1; __syncthreads(); marked = s_w[16] != 0; }
        if (!marked) { mc_group_sync<NT>(); return; }
    }
    uint32_t carry = 0;
    for (uint32_t k0 = 0; k0 < n; k0 += NT) {
        const uint32_t k = k0 + (uint32_t)tid;
        const bool head = k < n && (k == 0 || (x[k] >> 49) != (x[k - 1] >> 49) || MC_PLACE_OF(y[k]) != MC_PLACE_OF(y[k - 1]));
        const unsigned long long bal = __ballot(head);
        uint32_t base = 0, tot = (uint32_t)__popcll(bal);
        if (NT > 64) {
            if (lane == 0) s_w[wv] = tot;
            __syncthreads();
            tot = 0;
            for (int w = 0; w < NT / 64; w++) { const uint32_t c = s_w[w]; if (w < wv) base += c; tot += c; }
            __syncthreads();
        }
        if (k < n) R[k] = carry + base + (uint32_t)__popcll(bal & ((2ull << lane) - 1ull));
        carry += tot;
    }
    if (tid == 0) { *nv = carry; *nrow = 1u; }
    mc_group_sync<NT>();
    for (uint32_t k = (uint32_t)tid; k < n; k += NT) {              // a thread per subject
        if (k != 0 && (x[k] >> 49) == (x[k - 1] >> 49)) continue;
        const uint64_t sx = x[k] >> 49;
        uint32_t ge = k + 1;
        while (ge < n && (x[ge] >> 49) == sx) ge++;
        const uint32_t r0 = (uint32_t)R[k], kg = (uint32_t)R[ge - 1] - r0 + 1, ob = r0 - 1;
        for (uint32_t j = k; j < ge;) {
            const uint32_t rj = (uint32_t)R[j];
            uint32_t bestj = j, j2 = j + 1;
            while (j2 < ge && (uint32_t)R[j2] == rj) { if (MC_SCORE_OF(y[j2]) > MC_SCORE_OF(y[bestj])) bestj = j2; j2++; }
            const uint32_t run = rj - r0, o = ob + kg - 1 - run;
            order[o] = slots[(uint32_t)(x[bestj] & 0x1FFFFFu)];
            gsz[o] = run == kg - 1 ? kg : 0u;
            j = j2;
        }
    }
    mc_group_sync<NT>();
}
// scratch: 12 64-bit words per HSP (the finishing kernels' tmp): a segment too long for the LDS is sorted there - blocks of CAP
// items in LDS first, the merge passes above them in global memory (buffers at 12 a and 12 a + 4 n, the run counters at 12 a + 8 n)
template <int NT, uint32_t CAP>
__global__ void __launch_bounds__(NT) k_order_heavy(const uint64_t *__restrict__ keys, const uint64_t *__restrict__ places, const uint32_t *__restrict__ slots, const uint32_t *__restrict__ heads,
                                                    const uint32_t *__restrict__ list, const uint32_t *__restrict__ nlist_p, uint32_t *take, const uint8_t *__restrict__ low, uint32_t *order, uint32_t *gsz, uint32_t *nv,
                                                    uint32_t *nrow_of, uint64_t *scratch)
{
    uint64_t *lds = (uint64_t *)mc_smem;                            // 2 x CAP items and CAP 16-bit counters (dynamic LDS)
    uint16_t *ldsR = (uint16_t *)(lds + 2 * CAP);
    __shared__ uint32_t s_w[17], s_e;
    const int tid = (int)threadIdx.x;
    const uint32_t nlist = *nlist_p;
    uint32_t sub = 0, e0 = 0;
    for (;;) {
        // the next read of the list, whoever is free takes it (their sizes differ by orders of magnitude: dealt out in turn, the
        // workgroup that met the longest ones finished long after the others)
        // (the many mid-sized reads eight at a time: an atomic on ONE counter runs at the memory side, 125 M/s for the whole GPU)
        uint32_t e = 0;
        if (NT == 64) {
            if ((sub & 7u) == 0) { if (tid == 0) e = atomicAdd(take, 8u); e0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)e); }
            e = e0 + (sub++ & 7u);
        } else { __syncthreads(); if (tid == 0) s_e = atomicAdd(take, 1u); __syncthreads(); e = s_e; }
        if (e >= nlist) break;
        const uint32_t r = list[e], a = heads[r], n = heads[r + 1] - a;
        const bool marked = low[r] != 0;
        const uint64_t *kk = keys + a;
        uint32_t m = 64;
        while (m < n) m <<= 1;
        if (m <= CAP) {
#pragma unroll 4
            for (uint32_t k = (uint32_t)tid; k < m; k += NT) lds[k] = MC_ITEM_OF(kk, k, n);
            mc_group_sync<NT>();
            uint64_t *x = mc_group_mergesort<NT>(lds, lds + CAP, m, tid);
            mc_order_heavy_out<NT>(x, x == lds ? lds + CAP : lds, ldsR, slots + a, places + a, n, marked, order + a, gsz + a, nrow_of + r, nv + r, tid, s_w);
        } else {
            uint64_t *g = scratch + 12 * (size_t)a, *g2 = g + 4 * (size_t)n;      // (m < 2 n)
            for (uint32_t b0 = 0; b0 < m; b0 += CAP) {
#pragma unroll 4
                for (uint32_t k = (uint32_t)tid; k < CAP; k += NT) lds[k] = MC_ITEM_OF(kk, b0 + k, n);
                mc_group_sync<NT>();
                uint64_t *x = mc_group_mergesort<NT>(lds, lds + CAP, CAP, tid);
                for (uint32_t k = (uint32_t)tid; k < CAP; k += NT) g[b0 + k] = x[k];
                mc_group_sync<NT>();
            }
            __threadfence_block();
            for (uint32_t w = CAP; w < m; w <<= 1) {
                mc_merge_pass<NT>(g, g2, m, w, tid);
                __threadfence_block();
                mc_group_sync<NT>();
                uint64_t *t = g; g = g2; g2 = t;
            }
            __threadfence_block();
            mc_order_heavy_out<NT>(g, g2, (uint32_t *)(scratch + 12 * (size_t)a + 8 * (size_t)n), slots + a, places + a, n, marked, order + a, gsz + a, nrow_of + r, nv + r, tid, s_w);
        }
    }
}
