// k_finish.h - stage D: per-read finishing (SumEvalue@0x408a50, PrintRes@0x409310, MergeRes@0x40e3b0, classify_reads
// microbe_census.py:432-460): a thread per read (k_finish) or a wave per read (k_finish_heavy, k_heap_lanes, k_heavy_rows), rows out.
#pragma once
#include "mc_hip_common.h"
#include "mc_heap_words.h"

#ifndef MC_FH_MIN
#define MC_FH_MIN 96
#endif
#ifndef MC_FH_MIN_BEST
#define MC_FH_MIN_BEST 32
#endif
// MC_FH_MIN: reads with more stacked HSPs than this are finished by a whole wave (k_finish_heavy); MC_FH_MIN_BEST: the same with
// best hits only, where few reads are finished and the longest thread of k_finish decides (per 1 M reads of 150 bp: 96 / 48 / 32 / 16
// -> finishing 2.36 / 2.64 / 2.65 / 3.80 ms with rows, 2.38 / 1.43 / 1.41 / 1.42 ms with best hits only)
#ifndef MC_FH_N1          // (a test builds the library with small arrays so that ordinary reads take the paths of the largest ones)
#define MC_FH_N1 512      // subjects / ranked HSPs a read may have in the first wave kernel (11 KB of LDS per wave) ...
#define MC_FH_N2 1280     // ... in the second (28 KB: five waves per CU; 2048 - 45 KB, three waves - finished 0.1 ms per 2 M reads later) ...
#define MC_FH_N3 6144     // ... and in the third (135 KB, one wave per CU), where anything larger is finished by lane 0 alone
#endif

// The reads that get a wave of their own (k_finish_heavy), collected before the finishing kernels start so that they can run
// beside the thread-per-read kernel on a second stream.  A read without a marked HSP prints nothing whatever its size.
// The reads with a marked HSP that a single thread finishes (k_finish) are listed by size class as well: a wave of k_finish
// then holds reads of similar size instead of one read of 90 HSPs among 63 idle lanes.
#define MC_LIGHT_CLASS(n) ((n) <= 4 ? 0 : (n) <= 16 ? 1 : (n) <= 48 ? 2 : 3)
__global__ void __launch_bounds__(256) k_heavy_lists(const uint32_t *__restrict__ nv, uint32_t nheads, uint32_t *nrow_of,
                                                     McBestHit *best_of, uint32_t *counters, uint32_t *heavy, uint32_t *light, uint32_t light_pitch, uint32_t fh_min,
                                                     uint32_t *heavy1, uint32_t *heavy2, uint32_t *heavy3)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    int cls = -1, hc = 0;                                        // -1 nothing to do, 0..3 light class, 4 heavy (hc: which of the three wave-per-read kernels)
    if (s < nheads) {
        const uint32_t any = nrow_of[s];                           // (k_order_*: the read is marked, nv[s] = the size of its stacks)
        if (!any) best_of[s].family = -1;
        else { const uint32_t n = nv[s]; cls = n > fh_min ? 4 : MC_LIGHT_CLASS(n); hc = n <= MC_FH_N1 ? 1 : n <= MC_FH_N2 ? 2 : 3; }
    }
    // the heavy reads, and by the kernel whose arrays hold their stacks - known exactly now that the stacks are built before the finishing:
    // the three kernels run side by side (they used to hand the reads that did not fit from one to the next); the light reads by size class
    const int idx[8] = {C_HEAVY, C_HEAVY1, C_HEAVY2, C_HEAVY3, C_LIGHT0, C_LIGHT1, C_LIGHT2, C_LIGHT3};
    const uint32_t wants = cls == 4 ? (1u | (1u << hc)) : cls >= 0 ? (1u << (4 + cls)) : 0u;
    uint32_t off[8];
    mc_block_alloc_multi<8>(counters, idx, wants, off);
    if (cls == 4) {
        heavy[off[0]] = s;
        (hc == 1 ? heavy1 : hc == 2 ? heavy2 : heavy3)[off[hc]] = off[0];
    } else if (cls >= 0) light[(size_t)cls * light_pitch + off[4 + cls]] = s;
}

// The list of one wave-per-read kernel, LONGEST STACKS FIRST (a counting sort by one workgroup, as k_heap_order).  The kernel's waves
// take the list in turn (wave b: entries b, b + G, ...): in the order the atomics of k_heavy_lists left, the read that takes longest
// of all - half a millisecond against a launch of 0.6 - 0.7 - could be the second or third of its wave; the launches' durations
// moved by 30 % from box to box with it.
__global__ void __launch_bounds__(1024) k_heavy_order(const uint32_t *__restrict__ list, const uint32_t *__restrict__ count_p, const uint32_t *__restrict__ heavy, const uint32_t *__restrict__ nv,
                                                      int shift, uint32_t *out)
{
    __shared__ uint32_t bin[512], sc[512];
    const uint32_t n = *count_p, t = threadIdx.x;
    if (t < 512) bin[t] = 0;
    __syncthreads();
    auto key = [&](uint32_t i) -> uint32_t { const uint32_t k = nv[heavy[list[i]] & 0x7FFFFFFFu] >> shift; return 511u - (k > 511u ? 511u : k); };   // (bin 0: the longest)
    for (uint32_t i = t; i < n; i += 1024) atomicAdd(&bin[key(i)], 1u);
    __syncthreads();
    const uint32_t mine = t < 512 ? bin[t] : 0u;
    if (t < 512) sc[t] = mine;
    __syncthreads();
    for (uint32_t d = 1; d < 512; d <<= 1) { const uint32_t y = (t < 512 && t >= d) ? sc[t - d] : 0u; __syncthreads(); if (t < 512) sc[t] += y; __syncthreads(); }
    if (t < 512) bin[t] = sc[t] - mine;
    __syncthreads();
    for (uint32_t i = t; i < n; i += 1024) out[atomicAdd(&bin[key(i)], 1u)] = list[i];
}

// One thread per marked read.  All scratch is addressed by the read's offset into the binned HSPs (heads; a read never produces
// more rows than it has HSPs): v = the stacks (built by the ordering kernels), tmp = 2 HSP slots per HSP for the
// sum statistics, reused afterwards for the read's rows and their merge keys (64 + 8 bytes per row <= 96).  The rows
// stay in that scratch; k_emit_rows moves them to their final place once the row counts have been scanned.
// The (log E, index) items both order-defining sorts work on - std::sort of PrintRes, MergeRes' heap sort: some thousand dependent
// accesses for a read of 96 stacked HSPs - lie in the thread's own stretch of LDS (round 4; in global scratch every one of those
// accesses was a trip to the L2 / HBM, and the longest thread of the launch was the launch: 2.1 ms per 1 M reads whether 10,000 or
// 100,000 reads were finished).  TPB threads per workgroup, ITEMS items per thread; stretches are 4 words apart modulo the 32 banks
// (a wave reading 16 bytes per lane then meets the minimum of conflicts).  CL0: the first of the two size classes of the launch.
template <int TPB, int ITEMS, int CL0>
__global__ void __launch_bounds__(TPB) k_finish(const McTables *__restrict__ T, McIndex X, const McClassPars *__restrict__ P, const int32_t *__restrict__ fam,
                                                const uint32_t *__restrict__ nv, const uint32_t *__restrict__ heads, uint32_t nheads,
                                                McHsp *v, McHsp *tmp, int64_t first_read_id, uint32_t *nrow_of, McBestHit *best,
                                                const uint32_t *__restrict__ light, uint32_t light_pitch, const uint32_t *__restrict__ nlight, int use_lds)
{
    // blockIdx.y = size class, the larger first: two classes in ONE launch (and the two launches side by side on two streams) - a
    // thread finishes its read alone and the reads that print anything fill a fraction of the GPU
    constexpr int STRIDE = ITEMS * 16 + 16;
    const int cl = CL0 + 1 - (int)blockIdx.y;
    const uint32_t idx = blockIdx.x * TPB + threadIdx.x;
    if (idx >= nlight[cl]) return;                                // (the reads of one size class that have something to print: k_heavy_lists)
    const uint32_t s = light[(size_t)cl * light_pitch + idx];     // the read; its stacks: v[heads[s] ...], nv[s] records (k_order_*)
    const uint32_t a = heads[s];
    const int n = (int)(heads[s + 1] - a), vn = (int)nv[s];       // (the scratch of a read is laid out by the size of its segment)
    McRow *myrows = (McRow *)(tmp + 2 * (size_t)a);
    double *myk = (double *)(myrows + n);
    McBestHit bh;
    int nr;
    if (use_lds && vn <= ITEMS) nr = mc_finish_stacked_t<McSortsInl>(*T, X, *P, fam, (int)((int64_t)s + first_read_id), v + a, vn, tmp + 2 * (size_t)a, myrows, myk, (McSortItem *)(mc_smem + (size_t)threadIdx.x * STRIDE), &bh);
    else nr = mc_finish_stacked(*T, X, *P, fam, (int)((int64_t)s + first_read_id), v + a, vn, tmp + 2 * (size_t)a, myrows, myk, (McSortItem *)(myk + n), &bh);   // (64 n + 8 n + 16 n = 88 n <= 96 n bytes of the read's tmp area)
    nrow_of[s] = (uint32_t)nr;
    best[s] = bh;                                                 // per read that has HSPs (family -1: none); k_emit_rows collects them
}
// ---- std::sort (libstdc++ 4.4 introsort) replayed by a whole wave -----------------------------------------------------------
// mc_std_sort (mc_sort_impl.h) is the move-for-move statement; this computes the same permutation with the 64 lanes:
//  * __unguarded_partition: the left scan stops at the elements that are not < pivot, in order of position (A_0 < A_1 < ...),
//    the right scan at the elements that are not > pivot, from the right (B_0 > B_1 > ...); both scans only ever see elements
//    nobody has moved yet, so the k-th swap exchanges A_k and B_k as long as A_k < B_k, and the cut is min(A_K, B_(K-1)) for the
//    first K that fails (the swapped-in element at B_(K-1) stops the left scan at the latest).  Ranks by ballot + popcount,
//    all swaps at once.
//  * the recursion (depth limit, ranges of <= 16 left alone, heap-sort fallback by lane 0) is the reference's own;
//  * __final_insertion_sort is a stable sort of an array in which no element is further than 15 positions from its place
//    (ranges of <= 16 between ordered neighbours): place = position - (larger keys among the 15 before) + (smaller keys among
//    the 15 behind).
// Checked against mc_std_sort on 200,000 random arrays (ties, sorted, reversed; tests/test_emul.py runs the same formulation).
__device__ __forceinline__ void mc_wave_std_sort(McSortItem *items, int n, uint16_t *posA, uint16_t *posB, uint16_t *npos, int *stk, int lane)
{
    if (n <= 1) return;
    const unsigned long long lt = (1ull << lane) - 1;
    int lg = 0;
    for (int t = n; t > 1; t >>= 1) lg++;
    int sp = 1;
    if (lane == 0) { stk[0] = 0; stk[1] = n; stk[2] = 2 * lg; }
    mc_wave_sync();
    while (sp > 0) {
        sp--;
        int f = stk[3 * sp], l = stk[3 * sp + 1], depth = stk[3 * sp + 2];
        mc_wave_sync();
        while (l - f > 16) {
            if (depth == 0) { if (lane == 0) mc_heapsort_inl(items + f, (long)(l - f), 0); mc_wave_sync(); break; }
            --depth;
            const double x = items[f].k, y = items[f + (l - f) / 2].k, z = items[l - 1].k;
            double p;
            if (x < y) { if (y < z) p = y; else if (x < z) p = z; else p = x; }
            else if (x < z) p = x;
            else if (y < z) p = z;
            else p = y;
            int nA = 0, nB = 0;
            for (int c0 = f; c0 < l; c0 += 64) {
                const int i = c0 + lane;
                const bool fa = i < l && !(items[i < l ? i : f].k < p);
                const unsigned long long m = __ballot(fa);
                if (fa) posA[nA + __popcll(m & lt)] = (uint16_t)i;
                nA += __popcll(m);
            }
            for (int c0 = l - 1; c0 >= f; c0 -= 64) {
                const int i = c0 - lane;
                const bool fb = i >= f && !(p < items[i >= f ? i : f].k);
                const unsigned long long m = __ballot(fb);
                if (fb) posB[nB + __popcll(m & lt)] = (uint16_t)i;
                nB += __popcll(m);
            }
            mc_wave_sync();
            const int mn = nA < nB ? nA : nB;
            int K = 0;
            for (int k0 = 0; k0 < mn; k0 += 64) {
                const int k = k0 + lane;
                const unsigned long long m = __ballot(k < mn && posA[k < mn ? k : 0] < posB[k < mn ? k : 0]);
                K += __popcll(m);
                if (m != ~0ull) break;
            }
            for (int k0 = 0; k0 < K; k0 += 64) {
                const int k = k0 + lane;
                if (k < K) { const int a = posA[k], b = posB[k]; const McSortItem t1 = items[a], t2 = items[b]; items[a] = t2; items[b] = t1; }
            }
            int split;
            if (K == 0) split = posA[0];
            else if (K < nA) { const int a = posA[K], b = posB[K - 1]; split = a < b ? a : b; }
            else split = posB[K - 1];
            mc_wave_sync();
            if (lane == 0) { stk[3 * sp] = split; stk[3 * sp + 1] = l; stk[3 * sp + 2] = depth; }
            sp++;
            l = split;
        }
        mc_wave_sync();
    }
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int x = c0 + lane;
        if (x < n) {
            const double kx = items[x].k;
            int np = x;
            const int y0 = x - 15 > 0 ? x - 15 : 0, y1 = x + 15 < n - 1 ? x + 15 : n - 1;
            for (int yy = y0; yy < x; yy++) np -= (items[yy].k > kx) ? 1 : 0;
            for (int yy = x + 1; yy <= y1; yy++) np += (items[yy].k < kx) ? 1 : 0;
            npos[x] = (uint16_t)np;
        }
    }
    mc_wave_sync();
    McSortItem cur = items[lane < n ? lane : 0];
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int nx = c0 + 64 + lane;
        const McSortItem nxt = items[nx < n ? nx : 0];            // the next 64 are in registers before anything of this round is written
        mc_wave_sync();
        if (c0 + lane < n) items[npos[c0 + lane]] = cur;
        cur = nxt;
        mc_wave_sync();
    }
}

// ---- MergeRes' heap sort (std::partial_sort over the whole range) on packed words ---------------------------------------------
// The rows arrive in ascending log E, so their printed keys are non-decreasing: a row's key is replaced by its dense rank
// (the number of distinct printed keys in front of it) and the heap runs on 32-bit words rank << 16 | position, element e in word
// e + 1 - the two children of a node then share one aligned 64-bit LDS read.  mc_heapsort (mc_sort_impl.h) move for move.
__device__ __forceinline__ void mc_heapw_adjust(uint32_t *hw, int hole, int len, uint32_t value)
{
    const int top = hole;
    int sc = hole;
    while (sc < (len - 1) / 2) {
        sc = 2 * (sc + 1);
        const uint2 ch = *(const uint2 *)(hw + sc);                // elements sc - 1 and sc
        uint32_t pick = ch.y;
        if ((ch.y >> 16) < (ch.x >> 16)) { sc--; pick = ch.x; }
        hw[hole + 1] = pick; hole = sc;
    }
    if ((len & 1) == 0 && sc == (len - 2) / 2) { sc = 2 * (sc + 1); hw[hole + 1] = hw[sc]; hole = sc - 1; }
    int parent = (hole - 1) / 2;
    while (hole > top && (hw[parent + 1] >> 16) < (value >> 16)) { hw[hole + 1] = hw[parent + 1]; hole = parent; parent = (hole - 1) / 2; }
    hw[hole + 1] = value;
}
__device__ __forceinline__ void mc_heapw_sort(uint32_t *hw, int n)
{
    if (n >= 2) for (int parent = (n - 2) / 2;; parent--) { mc_heapw_adjust(hw, parent, n, hw[parent + 1]); if (parent == 0) break; }
    for (int m = n; m > 1;) { m--; const uint32_t v = hw[m + 1]; hw[m + 1] = hw[1]; mc_heapw_adjust(hw, 0, m, v); }
}

// A read with many HSPs (one that really comes from a marker gene: hundreds of homologous subjects): one wave.
// Parallel over lanes: the per-subject stacks and sum statistics (mc_finish_group per subject), the (log E, index) items,
// std::sort by log E (mc_wave_std_sort), the rows and their classification.  Sequential, by lane 0 on packed words in LDS:
// MergeRes' heap sort by printed log E, which has to replay libstdc++'s exact sequence of moves.
// Same scratch layout and same results as k_finish.
#ifdef MC_EXP_TIMING
__device__ unsigned long long g_fh_acc[8], g_fh_cnt[8];
__device__ unsigned long long g_fh_worst[12];   // the slowest read of the launches: cycles << 20 | stacked HSPs (atomicMax), then its cycles by phase
#define MC_FH_TICK(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); if (lane == 0) { fh_acc_[fcat_] += now_ - flast_; fh_acc_[8 + fcat_] += 1; } flast_ = now_; fcat_ = (k); } while (0)
#else
#define MC_FH_TICK(k) do { } while (0)
#endif
// the heap words of a heavy read: in its own scratch, behind the place of its rows (at most n rows of 72 bytes; the area holds 96 n
// bytes and the words need 4 n + 8)
static_assert(sizeof(McRow) == 72 && sizeof(McHsp) == 48, "mc_heavy_words: rows of 72 bytes in an area of 2 x 48 bytes per HSP");
__device__ __forceinline__ uint32_t *mc_heavy_words(McHsp *tmp, uint32_t a, int n) { return (uint32_t *)((uint8_t *)(tmp + 2 * (size_t)a) + (size_t)sizeof(McRow) * n); }

// MergeRes' heap sort for the heavy reads, ONE LANE PER READ: the sort replays libstdc++'s exact sequence of moves and is a chain
// of dependent LDS accesses - as lane 0 of the read's own wave it was half of the heavy kernels' time (cycle counters), with 63
// lanes waiting; here 64 reads are replayed side by side.  Words transposed in LDS (word e of lane l at e * 64 + l: lanes on the
// same word never share a bank), 502 words per lane = 128.5 KB: one wave per CU.
// The heavy reads in descending order of their rows (a counting sort by one workgroup): a wave of k_heap_lanes takes as long as the
// longest of its 64 heap sorts, and the few thousand reads with 400 - 500 rows, spread over the list, sat in nearly every wave - two
// rounds of waves (one per CU) each as long as a 500-row sort.  Together they fill a fifth of the first round, and the rest is short.
__global__ void __launch_bounds__(1024) k_heap_order(const uint32_t *__restrict__ heavy_first, const uint32_t *__restrict__ nrow_of, const uint32_t *__restrict__ counters, uint32_t *order)
{
    __shared__ uint32_t bin[MC_MAX_M8 + 2];
    const uint32_t nheavy = counters[C_HEAVY];
    for (uint32_t i = threadIdx.x; i < MC_MAX_M8 + 2; i += 1024) bin[i] = 0;
    __syncthreads();
    auto key = [&](uint32_t slot) -> uint32_t {                   // rows of the read (0: finished already, or nothing to sort)
        const uint32_t e = heavy_first[slot];
        if (!(e & 0x80000000u)) return 0u;
        const uint32_t n = nrow_of[e & 0x7FFFFFFFu];
        return n > MC_MAX_M8 ? (uint32_t)MC_MAX_M8 : n;
    };
    for (uint32_t slot = threadIdx.x; slot < nheavy; slot += 1024) atomicAdd(&bin[key(slot)], 1u);
    __syncthreads();
    {   // bin[k]: the reads with k rows -> the place of the first of them, the largest k first (a scan over the 501 bins by 512 threads)
        __shared__ uint32_t sc[512];
        const uint32_t t = threadIdx.x, mine = t <= MC_MAX_M8 ? bin[MC_MAX_M8 - t] : 0u;
        if (t < 512) sc[t] = mine;
        __syncthreads();
        for (uint32_t d = 1; d < 512; d <<= 1) { const uint32_t y = (t < 512 && t >= d) ? sc[t - d] : 0u; __syncthreads(); if (t < 512) sc[t] += y; __syncthreads(); }
        if (t <= MC_MAX_M8) bin[MC_MAX_M8 - t] = sc[t] - mine;
    }
    __syncthreads();
    for (uint32_t slot = threadIdx.x; slot < nheavy; slot += 1024) order[atomicAdd(&bin[key(slot)], 1u)] = slot;
}
__global__ void __launch_bounds__(64) k_heap_lanes(const uint32_t *__restrict__ heads, uint32_t nheads, uint32_t nhsps, McHsp *tmp, const uint32_t *__restrict__ nrow_of,
                                                   const uint32_t *__restrict__ counters, const uint32_t *__restrict__ heavy_first, const uint32_t *__restrict__ order)
{
    uint32_t *lds = (uint32_t *)mc_smem;
    __shared__ uint32_t s_nr[64];
    __shared__ uint32_t *s_hw[64];
    const int lane = mc_lane();
    const uint32_t nheavy = counters[C_HEAVY];
    for (uint32_t slot0 = blockIdx.x * 64u; slot0 < nheavy; slot0 += gridDim.x * 64u) {
        int n = 0;
        {
            const uint32_t at = slot0 + (uint32_t)lane;
            uint32_t *hw = nullptr;
            if (at < nheavy) {
                const uint32_t e = heavy_first[order[at]];
                if (e & 0x80000000u) {
                    const uint32_t s = e & 0x7FFFFFFFu, a = heads[s], b = heads[s + 1];
                    n = (int)nrow_of[s];
                    hw = mc_heavy_words(tmp, a, (int)(b - a));
                }
            }
            if (n < 2) n = 0;                                       // (nothing to sort)
            s_nr[lane] = (uint32_t)n; s_hw[lane] = hw;
        }
        __syncthreads();
        for (int r = 0; r < 64; r++) {                               // the words of the 64 reads in, read by read (coalesced)
            const int nr = (int)s_nr[r];
            const uint32_t *hw = s_hw[r];
            for (int e = 1 + lane; e <= nr; e += 64) lds[(e << 6) + r] = hw[e];
        }
        __syncthreads();
        if (n >= 2) {   // mc_heapsort (mc_sort_impl.h) move for move, element e in word e + 1; the keys are the upper halves (mc_heap_words.h, checked on the host by tests/test_emul.py)
            struct Acc { uint32_t *lds; int lane; __device__ __forceinline__ uint32_t get(int e) const { return lds[((e + 1) << 6) + lane]; } __device__ __forceinline__ void set(int e, uint32_t v) { lds[((e + 1) << 6) + lane] = v; } } acc = { lds, lane };
            mc_heap_words_sort(acc, n);
        }
        __syncthreads();
        for (int r = 0; r < 64; r++) {
            const int nr = (int)s_nr[r];
            uint32_t *hw = s_hw[r];
            for (int e = 1 + lane; e <= nr; e += 64) hw[e] = lds[(e << 6) + r];
        }
        __syncthreads();
    }
}

// The rows of the heavy reads in their final order, and their classification: one wave per read.
__global__ void __launch_bounds__(64) k_heavy_rows(const McTables *__restrict__ T, McIndex X, const McClassPars *__restrict__ P, const int32_t *__restrict__ fam,
                                                   const uint32_t *__restrict__ heads, uint32_t nheads,
                                                   const McHsp *__restrict__ v, McHsp *tmp, int64_t first_read_id, const uint32_t *__restrict__ nrow_of, McBestHit *best_of,
                                                   const uint32_t *__restrict__ counters, const uint32_t *__restrict__ heavy_first)
{
    const int lane = mc_lane();
    const uint32_t nheavy = counters[C_HEAVY];
    for (uint32_t slot = blockIdx.x; slot < nheavy; slot += gridDim.x) {
        const uint32_t e = heavy_first[slot];
        if (!(e & 0x80000000u)) continue;                            // (finished by lane 0 of the last wave kernel)
        const uint32_t s = e & 0x7FFFFFFFu, a = heads[s], b = heads[s + 1];
        const int n = (int)(b - a), nrows = (int)nrow_of[s];
        const int read_id = (int)((int64_t)s + first_read_id);
        const uint32_t *hw = mc_heavy_words(tmp, a, n);
        McRow *myrows = (McRow *)(tmp + 2 * (size_t)a);
        double bbits = -1.0; int bidx = 0x7fffffff, bfam = -1, baln = 0, btl = 0;
        for (int i = lane; i < nrows; i += 64) {
            McRow r;
            mc_fill_row(*T, read_id, v[a + (hw[i + 1] & 0xFFFFu)], r);
            myrows[i] = r;
            const int f = fam[r.subject], tl = (int)(X.off[r.subject + 1] - X.off[r.subject]);
            if (mc_row_passes(*P, r, f, tl, r.frame) && (bfam < 0 || bbits < r.bits)) { bbits = r.bits; bidx = i; bfam = f; baln = r.alnlen; btl = tl; }
        }
        // classify_reads keeps the first row with the highest bit score: reduce (bits desc, row index asc) over the lanes
        for (int d = 32; d > 0; d >>= 1) {
            const double ob = __shfl_down(bbits, d);
            const int oi = __shfl_down(bidx, d), of = __shfl_down(bfam, d), oa = __shfl_down(baln, d), ot = __shfl_down(btl, d);
            if (of >= 0 && (bfam < 0 || ob > bbits || (ob == bbits && oi < bidx))) { bbits = ob; bidx = oi; bfam = of; baln = oa; btl = ot; }
        }
        if (lane == 0) {
            McBestHit bh; bh.read = read_id; bh.family = bfam; bh.aln = bfam >= 0 ? baln : 0; bh.target_len = bfam >= 0 ? btl : 0; bh.bits = bfam >= 0 ? bbits : 0.0;
            best_of[s] = bh;
        }
    }
}

template <int MAXN, int CTR, int CTR_NEXT>
__global__ void __launch_bounds__(64) k_finish_heavy(const McTables *__restrict__ T, McIndex X, const McClassPars *__restrict__ P, const int32_t *__restrict__ fam,
                                                     const uint32_t *__restrict__ nv, const uint32_t *__restrict__ heads, uint32_t nheads,
                                                     McHsp *v, McHsp *tmp, int64_t first_read_id, uint32_t *nrow_of, McBestHit *best_of, uint32_t *counters,
                                                     uint32_t *heavy_first, const uint32_t *__restrict__ list, uint32_t *list_next)
{
    McSortItem *items = (McSortItem *)mc_smem;                      // MAXN sort items, then three index arrays (dynamic LDS)
    uint16_t *gst = (uint16_t *)(items + MAXN), *gkept = gst + (MAXN + 2), *gofs = gkept + (MAXN + 2);
    __shared__ int s_vn, s_nrows;
    __shared__ int s_stk[3 * 64];
    const int lane = mc_lane();
    const unsigned long long lt = (1ull << lane) - 1;
    const uint32_t nheavy = counters[CTR];
#ifdef MC_EXP_TIMING
    __shared__ unsigned long long fh_acc_[16];
    if (lane < 16) fh_acc_[lane] = 0;
    __syncthreads();
    unsigned long long flast_ = __builtin_readcyclecounter(); int fcat_ = 7;   // 0 group starts 1 groups 2 scan, items 3 sort 4 threshold, ranks 5 heap sort 6 rows 7 other
#endif
    for (uint32_t bi = blockIdx.x; bi < nheavy; bi += gridDim.x) {
#ifdef MC_EXP_TIMING
        unsigned long long rd0_[8];
        for (int k = 0; k < 8; k++) rd0_[k] = fh_acc_[k];
        __syncthreads();
#endif
        MC_FH_TICK(0);
        const uint32_t slot = list[bi];                               // position in the list of all heavy reads (heavy_first)
        const uint32_t s = heavy_first[slot] & 0x7FFFFFFFu, a = heads[s], b = heads[s + 1];
        const int nseg = (int)(b - a);                              // (the read's scratch is laid out by the size of its segment)
        const int n = (int)nv[s];                                   // its stacks: v[a, a + n) (k_order_*: the first record of a subject's stack carries the stack's size in .read)
        McHsp *in = v + a;
        const int read_id = (int)((int64_t)s + first_read_id);
        // (a read with more stacked HSPs than this kernel's arrays hold moves on - before anything is changed: the sum statistics
        // below work in place)
        bool punt = n > MAXN || n > 65535;
        int vn = 0, ng = 0;
        if (!punt) {
            for (int i0 = 0; i0 < n; i0 += 64) {                     // subjects: the starts of their stacks
                const int i = i0 + lane;
                const bool st = i < n && in[i].read != 0u;
                const unsigned long long m = __ballot(st);
                if (st) gst[ng + __popcll(m & lt)] = (uint16_t)i;
                ng += __popcll(m);
            }
            if (lane == 0) gst[ng] = (uint16_t)n;
            __syncthreads();
            // per subject: sum statistics - results stay at the group's own offset of v
            MC_FH_TICK(1);
            for (int g = lane; g < ng; g += 64) {
                const int g0 = gst[g], k = gst[g + 1] - g0;
                int kept = k;
                if (k > 1) { const int sidx = in[g0].sidx; kept = mc_sum_evalue(*T, in + g0, 0, k, (int)(X.off[sidx + 1] - X.off[sidx]), tmp + 2 * ((size_t)a + g0)); }
                gkept[g] = (uint16_t)kept;
            }
            __syncthreads();
            MC_FH_TICK(2);
            // offsets of the groups in the sequence PrintRes sorts (exclusive scan of the kept counts)
            {
                int carry = 0;
                for (int g0 = 0; g0 < ng; g0 += 64) {
                    const int g = g0 + lane;
                    int x = g < ng ? gkept[g] : 0, incl = x;
                    for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(incl, d); if (lane >= d) incl += y; }
                    if (g < ng) gofs[g] = (uint16_t)(carry + incl - x);
                    carry += __shfl(incl, 63);
                }
                if (lane == 0) { gofs[ng] = (uint16_t)carry; s_vn = carry; }
            }
            __syncthreads();
            vn = s_vn;
        }
        if (punt) {
            if (CTR_NEXT >= 0) { if (lane == 0) list_next[atomicAdd(&counters[CTR_NEXT < 0 ? 0 : CTR_NEXT], 1u)] = slot; }
            else if (lane == 0) {                                   // larger than the largest arrays: lane 0 alone, everything in the read's global scratch
                McRow *myrows = (McRow *)(tmp + 2 * (size_t)a);
                double *myk = (double *)(myrows + nseg);
                McSortItem *myitems = (McSortItem *)(myk + nseg);
                McBestHit bh;
                nrow_of[s] = (uint32_t)mc_finish_stacked(*T, X, *P, fam, read_id, in, n, tmp + 2 * (size_t)a, myrows, myk, myitems, &bh);
                best_of[s] = bh;
            }
            __syncthreads();
            continue;
        }
        for (int g = lane; g < ng; g += 64) {
            const int g0 = gst[g], o = gofs[g], k = gkept[g];
            for (int j = 0; j < k; j++) { McSortItem it; it.k = v[a + g0 + j].loge; it.i = (uint32_t)(g0 + j); it.pad = 0; items[o + j] = it; }
        }
        __syncthreads();
        MC_FH_TICK(3);
        mc_wave_std_sort(items, vn, gst, gkept, gofs, s_stk, lane);   // std::sort by log E (PrintRes)
        __syncthreads();
        MC_FH_TICK(4);
        // rows: at most 500, log E below the threshold (the items are in ascending log E, so the test is monotone)
        {
            const int lim = vn < MC_MAX_M8 ? vn : MC_MAX_M8;
            int cnt = 0;
            for (int i0 = 0; i0 < lim; i0 += 64) {
                const int i = i0 + lane;
                const bool ok = i < lim && v[a + items[i < lim ? i : 0].i].loge < T->loge_thr;
                cnt += __popcll(__ballot(ok));
            }
            if (lane == 0) s_nrows = cnt;
        }
        __syncthreads();
        const int nrows = s_nrows;
        for (int i = lane; i < nrows; i += 64) items[i].k = mc_round6(v[a + items[i].i].loge);
        __syncthreads();
        {   // dense ranks of the printed keys -> heap words rank << 16 | index of the HSP in v, into the read's scratch behind the
            // place of its rows (the groups' scratch is dead by now): MergeRes' heap sort and the rows follow in k_heap_lanes and
            // k_heavy_rows
            uint32_t *ghw = mc_heavy_words(tmp, a, nseg);
            int carry = 0;
            for (int i0 = 0; i0 < nrows; i0 += 64) {
                const int i = i0 + lane;
                const bool nw = i < nrows && i > 0 && items[i].k != items[i - 1].k;
                const unsigned long long m = __ballot(nw);
                if (i < nrows) ghw[i + 1] = ((uint32_t)(carry + __popcll(m & (lt | (1ull << lane)))) << 16) | items[i].i;
                carry += __popcll(m);
            }
        }
        if (lane == 0) { nrow_of[s] = (uint32_t)nrows; heavy_first[slot] = s | 0x80000000u; }   // (the flag: heap sort and rows still to come)
        __syncthreads();
        MC_FH_TICK(7);
#ifdef MC_EXP_TIMING
        if (lane == 0) {   // the slowest read of the launch decides how long the launch lasts: which one, and where its time went
            unsigned long long tot = 0;
            for (int k = 0; k < 8; k++) tot += fh_acc_[k] - rd0_[k];
            const unsigned long long mine = (tot << 20) | (unsigned long long)(n & 0xFFFFF);
            if (atomicMax(&g_fh_worst[0], mine) < mine) for (int k = 0; k < 8; k++) g_fh_worst[1 + k] = fh_acc_[k] - rd0_[k];   // (racy among near-equal maxima: a development aid)
        }
        __syncthreads();
#endif
    }
#ifdef MC_EXP_TIMING
    MC_FH_TICK(7);
    __syncthreads();
    if (lane < 8) { atomicAdd(&g_fh_acc[lane], fh_acc_[lane]); atomicAdd(&g_fh_cnt[lane], fh_acc_[8 + lane]); }
#endif
}

// rows of read s -> rows[rowoff[s] ...]: the m8 order (ascending read, RAPsearch2's order inside a read); the best hits of the
// reads that have one are collected (any order: the host sorts them by read), the reads with rows counted
__global__ void __launch_bounds__(256) k_emit_rows(const uint32_t *__restrict__ heads, uint32_t nheads, const uint32_t *__restrict__ nrow_of, const uint32_t *__restrict__ rowoff,
                                                   const McHsp *__restrict__ tmp, McRow *__restrict__ rows, uint32_t cap_rows, const McBestHit *__restrict__ best_of, McBestHit *best,
                                                   uint32_t *counters, int copy_rows)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t nr = 0, cp_n = 0;
    const uint2 *cp_src = nullptr;
    uint2 *cp_dst = nullptr;
    McBestHit bh; bh.family = -1;
    if (s < nheads) {
        nr = nrow_of[s];
        bh = best_of[s];
        const uint32_t off = rowoff[s];
        if (s == nheads - 1) { counters[C_ROWS] = off + nr; if (copy_rows && off + nr > cap_rows) counters[C_OVERFLOW] = 4; }
        if (copy_rows && off + nr <= cap_rows && nr > 0) { cp_src = (const uint2 *)(tmp + 2 * (size_t)heads[s]); cp_dst = (uint2 *)(rows + off); cp_n = nr * (uint32_t)(sizeof(McRow) / 8); }
    }
    {   // the rows of the block's reads, read by read with all 256 threads (8 bytes each, coalesced: a row is 72 bytes) - one read in twelve prints
        // anything, 23 rows on average, and a thread copying its read's rows alone moved 64 bytes per turn
        __shared__ const uint2 *l_src[256];
        __shared__ uint2 *l_dst[256];
        __shared__ uint32_t l_n[256], l_cnt;
        if (threadIdx.x == 0) l_cnt = 0;
        __syncthreads();
        if (cp_n) { const uint32_t k = atomicAdd(&l_cnt, 1u); l_src[k] = cp_src; l_dst[k] = cp_dst; l_n[k] = cp_n; }   // (any order: the destinations are disjoint)
        __syncthreads();
        const uint32_t cnt = l_cnt;
        for (uint32_t k = 0; k < cnt; k++) {
            const uint2 *src = l_src[k];
            uint2 *dst = l_dst[k];
            const uint32_t n16 = l_n[k];
            for (uint32_t i = threadIdx.x; i < n16; i += 256) dst[i] = src[i];
        }
    }
    const int idx[2] = {C_SEGS, C_BEST};
    uint32_t off[2];
    mc_block_alloc_multi<2>(counters, idx, (nr > 0 ? 1u : 0u) | (bh.family >= 0 ? 2u : 0u), off);
    if (bh.family >= 0) best[off[1]] = bh;
}
