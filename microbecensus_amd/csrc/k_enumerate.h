// k_enumerate.h - stage A2: seed enumeration and index probing (CHashSearch::Searching@0x415050): k_enumerate, the generic kernel
// (any .info threshold, one thread per frame); k_enumerate_q, the position-parallel kernel the marker database runs (threshold 0);
// k_enumerate_count, its counting form (mc_set_counting: every probe the reference searches is searched and counted - no filters).
#pragma once
#include "mc_hip_common.h"

struct DevEmit {
    McSeedTask *tasks; uint32_t *counters; uint32_t cap; uint32_t read; int frame; const McIndex *X; uint32_t emitted;
    __device__ void operator()(int bucket, int nst, int cnt, int seedlen, int nkey, int pos, int phase)
    {
        emitted += (uint32_t)cnt;
        uint32_t base = atomicAdd(&counters[C_TASKS], (uint32_t)cnt);
        if (base + (uint32_t)cnt > cap) { counters[C_OVERFLOW] = 1; return; }
        uint32_t b0 = X->bstart[bucket];
        for (int i = 0; i < cnt; i++) {
            McSeedTask t;
            t.chrono = MC_CHRONO(frame, pos, phase, nst + i); t.posting = X->post[b0 + nst + i];
            const uint32_t o0 = X->off[t.posting >> 11], o1 = X->off[(t.posting >> 11) + 1], abs = o0 + (t.posting & 0x7ff);
            t.read = MC_TASK_READ(read, o1 - abs);
            t.seedlen_nkey = MC_TASK_W3(abs, seedlen, nkey);
            tasks[base + i] = t;
        }
    }
};

__global__ void __launch_bounds__(256) k_enumerate(const McTables *__restrict__ T, McIndex X, const uint8_t *__restrict__ frames, int FP, int L,
                                                   int64_t nreads, McSeedTask *tasks, uint32_t cap, uint32_t *counters, unsigned long long *stats)
{
    int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= nreads * 6) return;
    int64_t r = tid / 6;
    int f = (int)(tid - r * 6);
    int qlen = (L - f % 3) / 3;
    DevEmit e{tasks, counters, cap, (uint32_t)r, f, &X, 0u};
    McSeedCount sc; sc.lookups = 0; sc.keyprobes = 0; sc.tasks = 0;
    mc_enumerate_seeds(*T, X, frames + (r * 6 + f) * FP, qlen, e, &sc);
    if (stats) { atomicAdd(&stats[S_LOOKUPS], (unsigned long long)sc.lookups); atomicAdd(&stats[S_KEYPROBES], (unsigned long long)sc.keyprobes); atomicAdd(&stats[S_TASKS], (unsigned long long)e.emitted); }
}

// ------------------------------------------------------------------------------------------------
// k_enumerate_count: position-parallel seed probing for databases whose .info frequency threshold is 0 (the marker DB) - the
// COUNTING form (mc_set_counting; bench.py's roofline.reference_pattern): it searches every probe the reference's Searching
// searches, with the reference's binary searches where those decide the count, and reports bucket lookups and key reads.  It is
// rounds 2 - 4's seed kernel k_enumerate_t0 (one wave per read, queues drained per read) without its filters; the kernel of the
// product path is k_enumerate_q below, which shares the position pass, the queue items and the append with it.
//
// With threshold 0 the seed-length carry of Searching@0x415050 collapses (mc_enumerate_seeds documents the general
// rule): a position whose own bucket is non-empty always uses a 9-mer (or is skipped), and only positions with an
// EMPTY bucket look at `prev` - to decide from where the 10-mer validity check of the neighbourhood starts.  `prev`
// is 9 if the nearest earlier non-skipped position with a non-empty bucket found a matching 9-mer range, else 6.
// So a read is handled in two parallel passes: (0) the exact 9-mer probes and the 36 neighbourhood probes of every
// position whose neighbourhood does not depend on `prev`; (1) the few positions that do.
//
// One wave per read: positions, (position, wildcard offset) pairs and probes are compacted through per-wave LDS queues so that
// every stage runs on 64 items.  Seed hits are appended to slots from a prefix sum; one global atomic per 2048 slots.
// ------------------------------------------------------------------------------------------------
#define MC_EN_QCAP 128
#define MC_EN_NCHUNK(L) ((((L) / 3 - 6) + 63) / 64 > 0 ? (((L) / 3 - 6) + 63) / 64 : 1)
static_assert(6 * MC_EN_NCHUNK(3 * MC_MAXAA) * 64 <= 2048, "a deferred position is kept in 11 bits beside the wildcard filter's 4-bit answer");
#define MC_EN_ROW(FP) ((((FP) + 10 + 7) / 8) * 4)   // bytes of a frame's row of reduced-alphabet codes, two per byte, padded past the last seed's key
#define MC_EN_RAWB(FP) ((6 * (FP) + 255) / 256 * 256)   // the NEXT read's six frames as they lie in global memory, fetched straight into LDS while this read is searched
#define MC_EN_CN(x) ((x) > 6 ? (x) - 6 : 0)
#define MC_EN_NPOS(L) ((2 * (MC_EN_CN((L) / 3) + MC_EN_CN(((L) - 1) / 3) + MC_EN_CN(((L) - 2) / 3)) + 7) / 8 * 8)   // seed positions of a read's six frames (padded): what pre and dq can hold
#define MC_EN_WAVE_LDS(FP, L) ((size_t)6 * MC_EN_ROW(FP) + MC_EN_RAWB(FP) + (size_t)MC_EN_NPOS(L) * (8 + 2))
#define MC_EN_BLK 2048u                     // task slots a wave reserves at a time (one global atomic per block, not per append)
#define MC_EN_SHORT 4                      // seed-hit ranges up to this long are written by the lane that found them
#define MC_TASK_NONE 0xFFFFFFFFu            // read id of the padding entries that close a partly used block
struct McEnWave {
    uint32_t setter[6][6]; uint32_t hit[6][6]; uint32_t blk_base, blk_used;
    unsigned long long q[MC_EN_QCAP];       // probes that passed the bucket bitmap
    unsigned long long eq[MC_EN_QCAP];      // (position, group) pairs the wildcard filter answered yes for: ten probes each
#ifdef MC_EXP_TIMING
    unsigned long long tacc[6], tcnt[6];
#endif
    unsigned long long hq[MC_EN_QCAP];      // probes whose first-residue group is longer than 8 keys (binary search)
};

// item: bucket(20) | qk(16)<<20 | pos(8)<<36 | frame(3)<<44 | phase(6)<<47
// Appends the seed hits of one batch of probes (lane: cnt postings starting at posting index nst of its bucket).
// TAGGED (k_enumerate_q): the items of a batch belong to several reads of the wave's chunk - bits 57..60 of an item hold the read's number inside
// the chunk and `read` is the chunk's first read; nobody keeps hit flags (the positions that need them ask the index themselves).
template <class WT, bool TAGGED>
__device__ __forceinline__ uint32_t mc_en_append(const McIndex &X, unsigned long long item, int cnt, int nst, uint32_t start, uint32_t read, WT *W,
                                                 McSeedTask *tasks, uint32_t cap, uint32_t *counters, int lane)
{
    unsigned long long m = __ballot(cnt > 0);
    if (m == 0) return 0;
    const int pos = (int)((item >> 36) & 0xFF), frame = (int)((item >> 44) & 7), phase = (int)((item >> 47) & 63);
    if constexpr (!TAGGED) { if (phase == 0 && cnt > 0) atomicOr(&W->hit[frame][pos >> 5], 1u << (pos & 31)); }
    const uint32_t myread = TAGGED ? read + (uint32_t)((item >> 57) & 15u) : read;
    // slot of every lane's range: prefix sum of the counts over the lanes
    const uint32_t incl = mc_wave_scan_add((uint32_t)cnt);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63), excl = incl - (uint32_t)cnt;
    uint32_t base;
    if (total > MC_EN_BLK) {                     // rare: a long range, reserved directly
        base = 0;
        if (lane == 0) base = atomicAdd(&counters[C_TASKS], total);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (base + total > cap) { if (lane == 0) counters[C_OVERFLOW] = 1; return 0; }
    } else {
        uint32_t bb = W->blk_base, bu = W->blk_used;
        mc_wave_sync();
        if (bu + total > MC_EN_BLK) {
            for (uint32_t i = bu + lane; i < MC_EN_BLK; i += 64) tasks[bb + i].read = MC_TASK_NONE;
            uint32_t nb = 0;
            if (lane == 0) nb = atomicAdd(&counters[C_TASKS], MC_EN_BLK);
            nb = (uint32_t)__builtin_amdgcn_readfirstlane((int)nb);
            if (nb + MC_EN_BLK > cap) { if (lane == 0) { counters[C_OVERFLOW] = 1; W->blk_used = MC_EN_BLK; } return 0; }
            bb = nb; bu = 0;
        }
        base = bb + bu;
        if (lane == 0) { W->blk_base = bb; W->blk_used = bu + total; }
        mc_wave_sync();
    }
    // Short ranges (most: a 10-mer of an unrelated read matches one or two markers) are written by their own lanes, all at once.
    // Long ones are written by the whole wave, one range after the other: a conserved 10-mer occurs in hundreds of homologous
    // markers, and a lane that wrote such a range alone would keep the other 63 waiting.
    if (cnt > 0 && cnt <= MC_EN_SHORT) {
        const uint32_t sn = phase == 0 ? MC_TASK_W3(0, 9, 3) : MC_TASK_W3(0, 10, 4);
        if constexpr (TAGGED) {
            // k_enumerate_q writes the INDEX of a hit's posting (in X.post / X.post8), not the posting: the posting and the subject's offsets of
            // every hit were 150 of the 810 scattered lines a read of 150 bp cost this kernel, which is bound by such lines;
            // k_eval_seeds<true>, which is not, fetches posting, position in the residue array and rest of the subject in one 8-byte
            // load (MC_POST8), asked for a chunk ahead
#pragma unroll
            for (int i = 0; i < MC_EN_SHORT; i++)
                if (i < cnt) {
                    McSeedTask t;
                    t.read = myread; t.chrono = MC_CHRONO(frame, pos, phase, (uint32_t)nst + (uint32_t)i); t.posting = start + (uint32_t)nst + (uint32_t)i; t.seedlen_nkey = sn;
                    tasks[base + excl + (uint32_t)i] = t;
                }
        } else {
        uint32_t pst[MC_EN_SHORT];
        uint2 ofs[MC_EN_SHORT];   // the postings and the subjects' two offsets (one load) first, then the stores: a store between two loads orders them (the pointers may alias)
#pragma unroll
        for (int i = 0; i < MC_EN_SHORT; i++) pst[i] = X.post[start + (uint32_t)nst + (uint32_t)(i < cnt ? i : 0)];
#pragma unroll
        for (int i = 0; i < MC_EN_SHORT; i++) __builtin_memcpy(&ofs[i], X.off + (pst[i] >> 11), 8);
#pragma unroll
        for (int i = 0; i < MC_EN_SHORT; i++)
            if (i < cnt) {
                McSeedTask t;
                const uint32_t abs = ofs[i].x + (pst[i] & 0x7ffu);
                t.read = MC_TASK_READ(myread, ofs[i].y - abs); t.chrono = MC_CHRONO(frame, pos, phase, (uint32_t)nst + (uint32_t)i); t.posting = pst[i];
                t.seedlen_nkey = sn | abs;
                tasks[base + excl + (uint32_t)i] = t;
            }
        }
    }
    {   // the long ranges as ONE list of hits, 128 of them per turn whatever range they belong to: lane x finds the range it is in
        // (binary search over the running sums of the lanes, by permute), fetches that lane's fields and writes one hit.  Range
        // after range - a turn of the wave each, most of them shorter than the wave, the load of the posting and the store of the
        // hit of one range finished before the next began - took a quarter of the kernel (cycle counters).
        const bool lng = cnt > MC_EN_SHORT;
        if (__ballot(lng)) {
            const uint32_t lc = lng ? (uint32_t)cnt : 0u;
            const uint32_t lincl = mc_wave_scan_add(lc);
            const uint32_t ltot = (uint32_t)__builtin_amdgcn_readlane((int)lincl, 63);
            const uint32_t lexcl = lincl - lc, hi32 = (uint32_t)(item >> 32), from = start + (uint32_t)nst;
            for (uint32_t x0 = 0; x0 < ltot; x0 += 128) {
                uint32_t pst[2], slot[2], chr[2], snk[2], rd[2];
                bool in[2];
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const uint32_t x = x0 + 64u * (uint32_t)u + (uint32_t)lane;
                    int ol = 0;                                            // lanes whose running sum is <= x: the owner of hit x
#pragma unroll
                    for (int stp = 32; stp > 0; stp >>= 1) { const uint32_t v = (uint32_t)__shfl((int)lincl, ol + stp - 1); if (v <= x) ol += stp; }
                    ol &= 63;
                    const uint32_t i = x - (uint32_t)__shfl((int)lexcl, ol), oh = (uint32_t)__shfl((int)hi32, ol);
                    const uint32_t ofrom = (uint32_t)__shfl((int)from, ol), oex = (uint32_t)__shfl((int)excl, ol), onst = (uint32_t)__shfl(nst, ol);
                    in[u] = x < ltot;
                    const int p2 = (int)((oh >> 4) & 0xFF), f2 = (int)((oh >> 12) & 7), ph2 = (int)((oh >> 15) & 63);
                    pst[u] = TAGGED ? ofrom + i : X.post[in[u] ? ofrom + i : 0u];   // (TAGGED: the posting's index - see above)
                    slot[u] = base + oex + i; chr[u] = MC_CHRONO(f2, p2, ph2, onst + i); snk[u] = ph2 == 0 ? MC_TASK_W3(0, 9, 3) : MC_TASK_W3(0, 10, 4);
                    rd[u] = TAGGED ? read + ((oh >> 25) & 15u) : read;
                }
#pragma unroll
                for (int u = 0; u < 2; u++) if constexpr (!TAGGED) { uint2 of; __builtin_memcpy(&of, X.off + (pst[u] >> 11), 8); const uint32_t abs = of.x + (pst[u] & 0x7ffu); snk[u] |= abs; rd[u] = MC_TASK_READ(rd[u], of.y - abs); }
#pragma unroll
                for (int u = 0; u < 2; u++)
                    if (in[u]) {
                        McSeedTask t;
                        t.read = rd[u]; t.chrono = chr[u]; t.posting = pst[u]; t.seedlen_nkey = snk[u];
                        tasks[slot[u]] = t;
                    }
            }
        }
    }
    return (uint32_t)cnt;
}

// One batch of (up to 64) probes.  Returns per lane: key reads of the reference (bits 32..), seed hits (bits 8..31);
// bits 0..7 (uniform): the new fill of the heavy queue.
__device__ __forceinline__ unsigned long long mc_en_process(const McIndex &X, unsigned long long item, bool active, uint32_t read, McEnWave *W, int hn,
                                                         McSeedTask *tasks, uint32_t cap, uint32_t *counters, int lane
#ifdef MC_EXP_TIMING
                                                         , unsigned long long *tl_, int *tc_
#endif
                                                         )
{
    int cnt = 0, lb = 0;
    uint32_t start = 0, kp = 0;
    int c0 = 0;
    bool heavy = false;
    if (active) {
        const int bucket = (int)(item & 0xFFFFF);
        const uint32_t qk = (uint32_t)((item >> 20) & 0xFFFF);
        const McBucketRec *R = X.rec + bucket;
        const int k6 = (int)(qk >> 12);
        start = R->start; c0 = R->cum[k6];
        const int ns = (int)R->cum[k6 + 1] - c0;
        heavy = ns > 8;
        if (ns > 0 && !heavy) cnt = mc_group_range8(X.keys + start + c0, ns, qk, &lb);   // (the counting form wants the lower bound of an empty range too)
        if (!heavy) { const int n = R->cum[11]; kp = mc_bsearch_reads(n, c0 + lb) + (cnt > 0 ? mc_bsearch_reads(n, c0 + lb + cnt) : 0u); }
    }
    const unsigned long long hm = __ballot(heavy);
    if (hm) {
        if (heavy) W->hq[hn + __popcll(hm & ((1ull << lane) - 1))] = item;
        hn += __popcll(hm);
        mc_wave_sync();
    }
#ifdef MC_EXP_TIMING
    { const unsigned long long now_ = __builtin_readcyclecounter(); if (lane == 0) { W->tacc[2] += now_ - *tl_; W->tcnt[2]++; } *tl_ = now_; *tc_ = 1; }
#endif
    const uint32_t nt = mc_en_append<McEnWave, false>(X, item, cnt, c0 + lb, start, read, W, tasks, cap, counters, lane);
    return ((unsigned long long)kp << 32) | ((unsigned long long)nt << 8) | (unsigned long long)hn;
}

// One batch of probes whose group needs the binary searches.
__device__ __forceinline__ unsigned long long mc_en_heavy(const McIndex &X, unsigned long long item, bool active, uint32_t read, McEnWave *W,
                                                       McSeedTask *tasks, uint32_t cap, uint32_t *counters, int lane)
{
    int cnt = 0, lb = 0, c0 = 0;
    uint32_t start = 0, kp = 0;
    if (active) {
        const int bucket = (int)(item & 0xFFFFF);
        const uint32_t qk = (uint32_t)((item >> 20) & 0xFFFF);
        const McBucketRec *R = X.rec + bucket;
        const int k6 = (int)(qk >> 12);
        start = R->start; c0 = R->cum[k6];
        const int ns = (int)R->cum[k6 + 1] - c0;
        cnt = mc_group_range_bs(X.keys + start + c0, ns, qk, &lb);
        { const int n = R->cum[11]; kp = mc_bsearch_reads(n, c0 + lb) + (cnt > 0 ? mc_bsearch_reads(n, c0 + lb + cnt) : 0u); }
    }
    const uint32_t nt = mc_en_append<McEnWave, false>(X, item, cnt, c0 + lb, start, read, W, tasks, cap, counters, lane);
    return ((unsigned long long)kp << 32) | ((unsigned long long)nt << 8);
}

#ifdef MC_EXP_TIMING
#define MC_TICK(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); if (lane == 0) { W->tacc[tcat] += now_ - tlast; W->tcnt[tcat]++; } tlast = now_; tcat = (k); } while (0)
#else
#define MC_TICK(k) do { } while (0)
#endif
#ifdef MC_EN_WPE                          // (experiments: force an occupancy)
#define MC_EN_ATTR __attribute__((amdgpu_waves_per_eu(MC_EN_WPE, MC_EN_WPE)))
#else
#define MC_EN_ATTR __attribute__((amdgpu_waves_per_eu(6, 6)))   // 80 VGPRs: the 24 waves per CU of the launch (the allocator stops at 83 by itself)
#endif
template <int MC_EN_WAVES>
__global__ void MC_EN_ATTR __launch_bounds__(64 * MC_EN_WAVES) k_enumerate_count(const McTables *__restrict__ T, McIndex X, const uint32_t *__restrict__ bitmap,
                                                                   const uint8_t *__restrict__ frames, int FP, int L, int64_t nreads, McSeedTask *tasks,
                                                                   uint32_t cap, uint32_t *counters, unsigned long long *stats, uint32_t read0, int
#ifdef MC_EN_FRONT_ONLY
                                                                   , uint32_t *, uint32_t, uint32_t *
#endif
                                                                   )
{
    uint8_t *smem = mc_smem;
    uint8_t *grp = smem;                                                    // 32-byte group table
    const int lane = mc_lane(), wv = threadIdx.x >> 6;
    McEnWave *W = (McEnWave *)(smem + 64 + (size_t)wv * sizeof(McEnWave));
    uint8_t *fr_all = smem + 64 + (size_t)MC_EN_WAVES * sizeof(McEnWave);
    const int FPn = MC_EN_ROW(FP);
    const int ql0 = L / 3, ql1 = (L - 1) / 3, ql2 = (L - 2) / 3;           // frame lengths (frames f and f + 3 alike)
    const int cn0 = ql0 > 6 ? ql0 - 6 : 0, cn1 = ql1 > 6 ? ql1 - 6 : 0, cn2 = ql2 > 6 ? ql2 - 6 : 0;   // seed positions of the frames, and their running sums
    const int cum1 = cn0, cum2 = cum1 + cn1, cum3 = cum2 + cn2, cum4 = cum3 + cn0, cum5 = cum4 + cn1, cum6 = cum5 + cn2;
    const uint32_t rcp_fpn = (65536u + (uint32_t)FPn - 1u) / (uint32_t)FPn;   // i / FPn = (i * rcp_fpn) >> 16 for the i < 6 * FPn in use
    uint8_t *fr = fr_all + (size_t)wv * MC_EN_WAVE_LDS(FP, L);
    uint8_t *raw = fr + 6 * FPn;                                            // the next read's frames, on their way (global_load_lds)
    unsigned long long *pre = (unsigned long long *)(raw + MC_EN_RAWB(FP));   // the positions of the read that probe anything (at most MC_EN_NPOS)
    uint16_t *dq = (uint16_t *)(pre + MC_EN_NPOS(L));                       // positions whose neighbourhood waits for the exact probes' results
    if (threadIdx.x < 32) grp[threadIdx.x] = T->grp[threadIdx.x];
    if (lane == 0) { W->blk_base = 0; W->blk_used = MC_EN_BLK; }
    __syncthreads();
    McSeedCount sc; sc.lookups = 0; sc.keyprobes = 0; sc.tasks = 0;
    uint32_t n_exact = 0, n_wild = 0, n_pairs = 0, n_probes = 0;         // what this wave asked its structures (wave-uniform)
    const unsigned long long lt = (1ull << lane) - 1;
    const int64_t nw = (int64_t)gridDim.x * MC_EN_WAVES;
#ifdef MC_EXP_TIMING
    if (lane < 6) { W->tacc[lane] = 0; W->tcnt[lane] = 0; }
    unsigned long long tlast = __builtin_readcyclecounter(); int tcat = 0;   // 0 staging/other 1 heavy 2 process 3 push 4 setup 5 expand
#endif
    // The frames of a read come from HBM; with one wave per read that trip stood at the head of every read.  They are fetched
    // straight into LDS (no registers) one read ahead: issued when this read's codes have been staged, needed when it is done.
    const int nraw = 6 * FP / 4;                                            // dwords of a read's frames (FP is a multiple of 4)
#define MC_EN_FETCH(rr)                                                                                                          \
    do {                                                                                                                         \
        const uint32_t *gs_ = (const uint32_t *)(frames + (rr) * 6 * FP);                                                        \
        for (int i0_ = 0; i0_ < nraw; i0_ += 64)                                                                                 \
            if (i0_ + lane < nraw) __builtin_amdgcn_global_load_lds(gs_ + i0_ + lane, (uint32_t *)raw + i0_, 4, 0, 0);            \
    } while (0)
    // Which reads a wave takes.  Dealt out in turn (read w, w + nw, ...) a wave's 325 reads of a 2 M batch cost what they cost - a read of
    // a marker gene tens of times an ordinary one - and the launch lasted as long as its unluckiest wave.  So: chunks of MC_EN_CHUNK
    // consecutive reads, the first one by the wave's number, every further one from a counter (one atomic per chunk: 250 k per launch).
#ifndef MC_EN_CHUNK
#define MC_EN_CHUNK 16
#endif
    int64_t r = ((int64_t)blockIdx.x * MC_EN_WAVES + wv) * MC_EN_CHUNK;
    int left = MC_EN_CHUNK - 1;                                              // reads of the current chunk behind r
    uint32_t pend = 0;                                                       // lane 0: the number of the chunk after this one
    static_assert(MC_EN_CHUNK >= 2, "the next chunk is asked for while the last but one read of a chunk is searched");
    if (r < nreads) MC_EN_FETCH(r);
    while (r < nreads) {
        int qn = 0, hn = 0, en = 0;
        MC_TICK(0);
        __builtin_amdgcn_s_waitcnt(0x0F70);                                  // vmcnt(0): this read's frames have arrived
        mc_wave_sync();
        {   // stage the six frames of this read as reduced-alphabet codes, two per byte (rows of FPn bytes, padded with the
            // invalid code: a seed's key residues past the frame end then read as invalid by themselves); clear the flags
            const uint8_t *src = raw;
            for (int i = lane; i < 6 * FPn; i += 64) {                         // byte i of the six rows
                const int f = (int)(((uint32_t)i * rcp_fpn) >> 16), b2 = 2 * (i - f * FPn);
                uint32_t g0 = MC_INVGRP, g1 = MC_INVGRP;
                if (b2 < FP) { const uint32_t two = *(const uint16_t *)(src + f * FP + b2); g0 = grp[two & 0xFF]; g1 = grp[two >> 8]; }   // (FP is a multiple of 4)
                fr[i] = (uint8_t)(g0 | (g1 << 4));
            }
            if (lane < 36) { ((uint32_t *)W->setter)[lane] = 0; ((uint32_t *)W->hit)[lane] = 0; }
            mc_wave_sync();
        }
        int64_t rnext = r + 1;                                               // the read after this one: its frames start their way now
        if (left == 1 && lane == 0) pend = atomicAdd(&counters[C_ENCHUNK], 1u);   // (the next chunk, asked for one read ahead: the answer is there when it is needed)
        if (left > 0) left--;
        else { rnext = (nw + (int64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)pend)) * MC_EN_CHUNK; left = MC_EN_CHUNK - 1; }
        if (rnext < nreads) MC_EN_FETCH(rnext);
        // What a seed position will do is decided here, once: the 6-mer's bucket and the four key residues (ten codes = 40
        // bits out of three aligned words of the row), whether the bucket holds anything (bitmap gather; those of three sweeps
        // are in flight together), and from that which probes it makes.  The positions of the six frames are numbered through
        // (every sweep but the last has 64 of them), and only the positions that probe anything are kept - a third have an
        // invalid residue in the 6-mer or nothing to ask: entry = seed 20 | g6..g9 16 | position 8 | frame 3 (the four fields
        // of a queue item, in place) | exact probe 1 | neighbourhood 1 | neighbourhood decided in pass 1 1 | g3 g4 g5 12
        int npre = 0;                                    // positions kept
        for (int k0 = 0; k0 < cum6; k0 += 192) {
            uint32_t sdv[3], gkv[3], bw[3], d3v[3], pfv[3];
            bool vd[3];
#pragma unroll
            for (int u = 0; u < 3; u++) {
                sdv[u] = 0; gkv[u] = 0; vd[u] = false; d3v[u] = 0; pfv[u] = 0;
                if (k0 + u * 64 >= cum6) continue;                             // (uniform)
                const int flat = k0 + u * 64 + lane;
                const int f = (flat >= cum1) + (flat >= cum2) + (flat >= cum3) + (flat >= cum4) + (flat >= cum5);
                const int pos = flat - (f == 0 ? 0 : f == 1 ? cum1 : f == 2 ? cum2 : f == 3 ? cum3 : f == 4 ? cum4 : cum5);
                const uint32_t *rw = (const uint32_t *)(fr + f * FPn) + (pos >> 3);   // (past the last position: some words of the wave's LDS, not used)
                const int o4 = (pos & 7) * 4;
                const uint32_t w0 = rw[0], w1 = rw[1], w2 = rw[2];
                unsigned long long v = (((unsigned long long)w1 << 32) | w0) >> o4;
                if (o4 == 28) v |= (unsigned long long)w2 << 36;
                const uint32_t six = (uint32_t)v & 0xFFFFFFu, y = six ^ 0xAAAAAAu;
                const bool bad = ((y - 0x111111u) & ~y & 0x888888u) != 0;      // one of the six codes is the invalid one
                const uint32_t seed = (six & 15u) * 100000u + ((six >> 4) & 15u) * 10000u + ((six >> 8) & 15u) * 1000u + ((six >> 12) & 15u) * 100u + ((six >> 16) & 15u) * 10u + (six >> 20);
                const uint32_t hi4 = (uint32_t)(v >> 24) & 0xFFFFu;            // g6 lowest
                const uint32_t gk = ((hi4 & 15u) << 12) | (((hi4 >> 4) & 15u) << 8) | (((hi4 >> 8) & 15u) << 4) | (hi4 >> 12);
                const bool ok = flat < cum6 && !bad;
                sdv[u] = ok ? seed : 0u; gkv[u] = gk; vd[u] = ok; d3v[u] = (six >> 12) & 0xFFFu; pfv[u] = (uint32_t)pos | ((uint32_t)f << 8);
            }
#pragma unroll
            for (int u = 0; u < 3; u++) bw[u] = bitmap[sdv[u] >> 5];
#pragma unroll
            for (int u = 0; u < 3; u++) {
                if (k0 + u * 64 >= cum6) break;
                const int pos = (int)(pfv[u] & 0xFF), f = (int)(pfv[u] >> 8), fm = f >= 3 ? f - 3 : f;
                const int rest = (fm == 0 ? cn0 : fm == 1 ? cn1 : cn2) - pos;   // residues behind the 6-mer
                const bool occ = (bw[u] >> (sdv[u] & 31)) & 1u;
                const uint32_t gk = gkv[u];
                const bool v6 = (gk >> 12) != MC_INVGRP, v7 = ((gk >> 8) & 15u) != MC_INVGRP, v8 = ((gk >> 4) & 15u) != MC_INVGRP, v9 = (gk & 15u) != MC_INVGRP;
                const bool live0 = vd[u] && occ && rest >= 3 && v6 && v7;        // exact 9-mer probe: it also defines `prev` for the positions behind it
                // The neighbourhood's validity check starts at residue `used`: 9 when the own bucket is occupied, else 8 or 6
                // depending on whether the nearest earlier exact probe of the frame found a range (prev).  That only matters
                // when g8, g9 are valid and g6 or g7 is not: those few positions are decided in pass 1.
                bool live = false, defer = false;
                if (vd[u] && rest >= 4) { if (occ) live = v6 && v7 && v9; else if (v8 && v9) { if (v6 && v7) live = true; else defer = true; } }
                if (vd[u]) sc.lookups += live0 ? 2 : 1;               // bucket-size probe of the exact seed, and its key-range probe
                if (live0) atomicOr(&W->setter[f][pos >> 5], 1u << (pos & 31));
                const bool keep = live0 || live || defer;
                const unsigned long long km = __ballot(keep);
                if (keep) pre[npre + __popcll(km & lt)] = (unsigned long long)sdv[u] | ((unsigned long long)gk << 20) | ((unsigned long long)pfv[u] << 36) |
                                                          ((unsigned long long)(live0 ? 1u : 0u) << 47) | ((unsigned long long)(live ? 1u : 0u) << 48) | ((unsigned long long)(defer ? 1u : 0u) << 49) | ((unsigned long long)d3v[u] << 50);
                npre += __popcll(km);
            }
        }
        mc_wave_sync();
        // Per position: its exact 9-mer, and its one-substitution 10-mers in four groups of ten probes (groups 0..2 =
        // offsets 4, 5, 3 of the 6-mer: neighbour buckets; group 3 = offset 6: same bucket, first key residue substituted).
        // Pass 0 sweeps the kept positions 64 at a time and generates both.  Whether a position has a neighbourhood
        // depends, for a few of them (own bucket empty, g8 and g9 valid, g6 or g7 not: ~3 % of the positions), on whether
        // the nearest earlier exact probe of the frame found a range; those wait in the list dq until pass 0 has drained
        // its queues, and are generated in pass 1.
        //   exact 9-mer  -> queue q
        //   10-mers      -> queue eq of (position, group) pairs -> 64 pairs at a time: the bucket bitmap decides which of the ten
        //                   probes of a pair are searched -> queue q
        //   q            -> bucket records: group scan; long groups through queue hq to the reference's binary searches -> seed hits
        // The generator is a state machine so that each stage exists once in the kernel.
        int dn = 0;                                      // deferred positions (dq)
        for (int pass = 0; pass < 2; pass++) {
            int flat0 = 0, dpos = 0;
            bool more = true;
            uint32_t wm = 0, wdig = 0;                   // groups of this lane's position that still have to enter eq; their own residues at the wildcard offsets
            unsigned long long wbase = 0;                // seed | key | position | frame of this lane's position
            uint32_t pm = 0;                             // surviving probes of this lane's expanded pair ...
            unsigned long long xi = 0;                   // ... and the pair itself
            for (;;) {
                const bool pmz = __ballot(pm != 0) == 0, wmz = __ballot(wm != 0) == 0;
                const bool tail = !more && wmz && en == 0 && pmz;            // nothing more will enter q
                if (hn >= 64 || (tail && qn == 0 && hn > 0)) {           // probes that need the binary searches
                    MC_TICK(1);
                    const int take = hn < 64 ? hn : 64;
                    hn -= take;
                    const unsigned long long rh = mc_en_heavy(X, (lane < take) ? W->hq[hn + lane] : 0ull, lane < take, read0 + (uint32_t)r, W, tasks, cap, counters, lane);
                    sc.keyprobes += (uint32_t)(rh >> 32); sc.tasks += (uint32_t)(rh >> 8) & 0xFFFFFFu;
                    mc_wave_sync();
                    continue;
                }
                if (qn >= 64 || (tail && qn > 0)) {                      // probes that passed the filters
                    MC_TICK(2);
                    const int take = qn < 64 ? qn : 64;
                    qn -= take;
                    n_probes += (uint32_t)take;
                    const unsigned long long ret = mc_en_process(X, (lane < take) ? W->q[qn + lane] : 0ull, lane < take, read0 + (uint32_t)r, W, hn, tasks, cap, counters, lane
#ifdef MC_EXP_TIMING
                                                                             , &tlast, &tcat
#endif
                                                                             );
                    hn = __builtin_amdgcn_readfirstlane((int)(ret & 0xFF));
                    sc.keyprobes += (uint32_t)(ret >> 32); sc.tasks += (uint32_t)(ret >> 8) & 0xFFFFFFu;
                    mc_wave_sync();
                    continue;
                }
                if (!pmz) {                                              // queue the surviving probes: one per lane and turn, until q holds a full batch
                    MC_TICK(3);
                    const int gc = (int)((xi >> 47) & 3), sd = (int)(xi & 0xFFFFF);
                    const uint32_t xk = (uint32_t)((xi >> 20) & 0xFFFF);
                    const int st = gc == 0 ? 10 : gc == 1 ? 1 : gc == 2 ? 100 : 0;
                    const int dd = (int)((xi >> 53) & 15);
                    const int s0 = sd - dd * st;                         // the bucket with the substituted digit taken out (gc 3: the bucket itself)
                    const unsigned long long keep = xi & 0x00007FF000000000ull;
                    for (;;) {
                        const unsigned long long pmm = __ballot(pm != 0);
                        if (pmm == 0 || qn >= 64) break;
                        const int j = __builtin_ctz(pm | 0x400u);
                        const int v = s0 + j * st;
                        const uint32_t k2 = gc < 3 ? xk : ((xk & 0x0FFFu) | ((uint32_t)j << 12));
                        if (pm) W->q[qn + __popcll(pmm & lt)] = keep | (unsigned long long)v | ((unsigned long long)k2 << 20) | ((unsigned long long)(1 + gc * 10 + j) << 47);
                        qn += __popcll(pmm);
                        pm &= pm - 1;
                    }
                    mc_wave_sync();
                    continue;
                }
                if (en >= 64 || (!more && wmz && en > 0)) {              // expand 64 (position, group) pairs into their ten probes
                    MC_TICK(5);
                    const int take = en < 64 ? en : 64;
                    en -= take;
                    n_pairs += (uint32_t)take;
                    const bool act = lane < take;
                    xi = act ? W->eq[en + lane] : 0ull;
                    const int gl = (int)((xi >> 47) & 3), sd = (int)(xi & 0xFFFFF);
                    const int st = gl == 0 ? 10 : gl == 1 ? 1 : gl == 2 ? 100 : 0;
                    const int d = (int)((xi >> 53) & 15);                // the position's own residue at the wildcard offset
                    uint32_t ok = 0;
#pragma unroll
                    for (int j = 0; j < 10; j++) {
                        const int v = sd + (j - d) * st;                 // st = 0 for the key group: the bucket stays
                        bool c = act && j != d;
                        if (c) { sc.lookups++; c = (bitmap[v >> 5] >> (v & 31)) & 1; }   // bucket occupancy decides, then the search
                        ok |= (uint32_t)c << j;
                    }
                    pm = ok;
                    mc_wave_sync();
                    continue;
                }
                if (!wmz) {                                              // pending groups enter eq: one per lane and turn, until eq holds a full batch
                    MC_TICK(3);
                    for (;;) {
                        const unsigned long long wmm = __ballot(wm != 0);
                        if (wmm == 0 || en >= 64) break;
                        const int gl = __builtin_ctz(wm | 16u);
                        if (wm) W->eq[en + __popcll(wmm & lt)] = wbase | ((unsigned long long)gl << 47) | ((unsigned long long)((wdig >> (4 * gl)) & 15u) << 53);
                        en += __popcll(wmm);
                        wm &= wm - 1;
                    }
                    mc_wave_sync();
                    continue;
                }
                if (!more) break;
                {   // next 64 positions: of the kept ones (pass 0) or of the deferred list (pass 1)
                    MC_TICK(4);
                    int idx;
                    bool here;
                    uint32_t wmd = 0;
                    if (pass == 0) {
                        if (flat0 >= npre) { more = false; continue; }
                        idx = flat0 + lane;
                        flat0 += 64;
                        here = idx < npre;
                    } else {
                        if (dpos >= dn) { more = false; continue; }
                        here = dpos + lane < dn;
                        const uint32_t e = here ? dq[dpos + lane] : 0u;      // position | the wildcard filter's answer, asked in pass 0
                        dpos += 64;
                        idx = (int)(e & 0x7FFu); wmd = e >> 11;
                    }
                    const unsigned long long pw = pre[here ? idx : 0];
                    const uint32_t qk = (uint32_t)(pw >> 20) & 0xFFFFu;
                    const uint32_t d3 = (uint32_t)(pw >> 50) & 15u, d4 = (uint32_t)(pw >> 54) & 15u, d5 = (uint32_t)(pw >> 58) & 15u;   // bucket digits at offsets 3, 4, 5
                    wdig = d4 | (d5 << 4) | (d3 << 8) | ((qk >> 12) << 12);  // the residue at the wildcard offset of groups 0..3
                    wbase = pw & 0x00007FFFFFFFFFFFull;                      // seed | key | position | frame: a queue item without its phase
                    if (pass == 0) {
                        const bool live0 = here && ((pw >> 47) & 1), live = here && ((pw >> 48) & 1);
                        bool defer = here && ((pw >> 49) & 1);
                        const unsigned long long m9 = __ballot(live0), mw = __ballot(live || defer);
                        n_exact += (uint32_t)__popcll(m9); n_wild += (uint32_t)__popcll(mw);
                        const bool pr = live0;                           // (no filters: every probe the reference searches is searched)
                        const unsigned long long prm = __ballot(pr);
                        if (prm) {
                            if (pr) W->q[qn + __popcll(prm & lt)] = wbase | (0xFull << 20);   // phase 0; key g6 g7 g8 F
                            qn += __popcll(prm);
                        }
                        const uint32_t wmt = 0xFu;                       // every probe is generated and searched
                        wm = live ? wmt : 0u;
                        defer = defer && wmt != 0;                       // (no group can match: nothing to decide later)
                        const unsigned long long dm = __ballot(defer);
                        if (dm) { if (defer) dq[dn + __popcll(dm & lt)] = (uint16_t)((uint32_t)idx | (wmt << 11)); dn += __popcll(dm); }
                    } else {   // a deferred position: own bucket empty, g8 and g9 valid, g6 or g7 invalid -> live iff prev == 9
                        bool live = false;
                        if (here) {
                            const int pos = (int)((pw >> 36) & 0xFF), fl = (int)((pw >> 44) & 7);
                            int w = pos >> 5;
                            uint32_t m = W->setter[fl][w] & ((1u << (pos & 31)) - 1);
                            while (m == 0 && w > 0) { w--; m = W->setter[fl][w]; }
                            if (m) { const int bb = 31 - __builtin_clz(m); live = (W->hit[fl][w] >> bb) & 1; }
                        }
                        wm = live ? wmd : 0u;
                    }
                    for (;;) {   // the groups enter eq at once while it has room (else from the state above)
                        const unsigned long long wmm = __ballot(wm != 0);
                        if (wmm == 0 || en >= 64) break;
                        const int gl = __builtin_ctz(wm | 16u);
                        if (wm) W->eq[en + __popcll(wmm & lt)] = wbase | ((unsigned long long)gl << 47) | ((unsigned long long)((wdig >> (4 * gl)) & 15u) << 53);
                        en += __popcll(wmm);
                        wm &= wm - 1;
                    }
                    mc_wave_sync();
                }
            }
            mc_wave_sync();
        }
        r = rnext;
    }
    MC_TICK(0);
#ifdef MC_EXP_TIMING
    mc_wave_sync();
    if (lane == 0) for (int k = 0; k < 6; k++) { atomicAdd(&stats[4 + k], W->tacc[k]); atomicAdd(&stats[10 + k], W->tcnt[k]); }
#endif
    {   // close the wave's last block
        mc_wave_sync();
        const uint32_t bb = W->blk_base, bu = W->blk_used;
        for (uint32_t i = bu + lane; i < MC_EN_BLK; i += 64) tasks[bb + i].read = MC_TASK_NONE;
    }
    {
        unsigned long long a = sc.lookups, b = sc.keyprobes, c = sc.tasks;
        for (int d = 32; d > 0; d >>= 1) { a += __shfl_down(a, d); b += __shfl_down(b, d); c += __shfl_down(c, d); }
        if (lane == 0) { atomicAdd(&stats[S_LOOKUPS], a); atomicAdd(&stats[S_KEYPROBES], b); atomicAdd(&stats[S_TASKS], c); atomicAdd(&stats[S_EXACT], (unsigned long long)n_exact); atomicAdd(&stats[S_WILD], (unsigned long long)n_wild); atomicAdd(&stats[S_PAIRS], (unsigned long long)n_pairs); atomicAdd(&stats[S_PROBES], (unsigned long long)n_probes); }
    }
}

// ------------------------------------------------------------------------------------------------
// k_enumerate_q (round 5): k_enumerate_t0 with queues that PERSIST ACROSS THE READS of a wave's chunk.
//
// k_enumerate_t0 drains its queues at the end of every read: a read of 150 bp has 200 kept positions, 92 (position, offset) pairs
// and 52 probes - against batches of 64 that is 3.1 sweeps, 1.4 expansions and 0.8 probe batches with 78 %, 72 % and 81 % of the
// lanes (47 of 64 active lanes by the counters), and the kernel is bound by VALU issue.  Here a wave owns the
// 16 consecutive reads of a chunk as ONE stream of work: the kept positions of a read are appended to the list `pre` behind what
// the reads before left over, a sweep takes 64 of them whatever reads they belong to, and so on down the pipeline - every item
// carries its read's number inside the chunk (4 bits: 57..60 of a queue item, a byte beside a `pre` entry).  The queues are
// drained once per chunk (one partial batch per stage and 16 reads).  What made the per-read drain necessary in k_enumerate_t0 is
// solved differently: the few positions whose neighbourhood depends on whether the nearest earlier exact probe of their frame
// found a range ("deferred": own bucket empty, g8 g9 valid, g6 or g7 not; ~3 % of the positions) no longer wait for that probe's
// result in per-read hit flags - they carry that probe's own (bucket, key) along (read off the frame's row while it is still
// staged) and ASK THE INDEX THEMSELVES, 64 of them at a time, in a stage of their own: 9-mer filter word and wildcard line
// together, then bucket record and key group / range table for those the filter lets through.
//
// Per wave (LDS): q, eq (128 items each), dq + dk (128 deferred positions: own item + the earlier probe's bucket and key),
// pre + ptag (256 kept positions: a slice of 192 positions is decoded whenever fewer than 64 are left), the setter flags and
// the staged rows of the CURRENT read, the raw frames of the NEXT one (global_load_lds).  At 150 bp 6.6 KB per wave - 24 waves
// per CU as before -, and no part of it grows with the read length but rows and raw frames (k_enumerate_t0 kept a whole read's
// positions: 16 waves per CU at 300 bp, 20 here).
// ------------------------------------------------------------------------------------------------
#define MC_ENQ_PCAP 256
#define MC_ENQ_DCAP 128
#define MC_ENQ_SLICE 192                   // positions decoded per visit of the position stage (three sub-slices whose bitmap gathers are in flight together)
static_assert(MC_EN_CHUNK == 16, "an item's read is 4 bits: its number inside a chunk of 16 reads that starts at a multiple of 16");
static_assert(63 + MC_ENQ_SLICE <= MC_ENQ_PCAP, "a slice is decoded when fewer than 64 positions are left in pre");
static_assert(63 + MC_ENQ_SLICE / 4 + 12 <= MC_ENQ_DCAP, "deferred positions: at most two per eight positions of a frame and two per frame boundary in a slice");
struct McEnqWave {
    uint32_t blk_base, blk_used;
    unsigned long long q[MC_EN_QCAP];       // probes that passed the filters
    unsigned long long eq[MC_EN_QCAP];      // (position, group) pairs the wildcard filter answered yes for
    unsigned long long dq[MC_ENQ_DCAP];     // deferred positions: seed 20 | g6..g9 16 | position 8 | frame 3 | read 4 << 57 (a queue item without its phase)
    unsigned long long pre[MC_ENQ_PCAP];    // kept positions (the entry of k_enumerate_t0 without its defer bit)
    uint32_t dk[MC_ENQ_DCAP];               // ... and the exact probe they depend on: bucket 20 | g6 g7 g8 12 (0xFFFFFFFF: none - the position has no neighbourhood)
    uint8_t ptag[MC_ENQ_PCAP];              // the read (number inside the chunk) of a pre entry
};
#define MC_ENQ_SETW(L) ((6 * ((MC_EN_CN((L) / 3) + 31) / 32) + 1) / 2 * 2)   // words of the current read's setter flags (frame f: words f * nw ..)
#define MC_ENQ_WAVE_LDS(FP, L) (sizeof(McEnqWave) + (size_t)MC_ENQ_SETW(L) * 4 + (size_t)6 * MC_EN_ROW(FP) + MC_EN_RAWB(FP))
static_assert(sizeof(McEnqWave) % 8 == 0, "the per-wave blocks stay 8-byte aligned");

// one batch of (up to 64) probes of the reads of the chunk that starts at read rbase: group scan or range table, then the append
__device__ __forceinline__ uint32_t mc_enq_process(const McIndex &X, unsigned long long item, bool active, uint32_t rbase, McEnqWave *W, McSeedTask *tasks, uint32_t cap,
                                                   uint32_t *counters, int lane)
{
    int cnt = 0, lb = 0, c0 = 0;
    uint32_t start = 0;
    if (active) {
        const int bucket = (int)(item & 0xFFFFF);
        const uint32_t qk = (uint32_t)((item >> 20) & 0xFFFF);
        const McBucketRec *R = X.rec + bucket;
        const int k6 = (int)(qk >> 12);
        start = R->start; c0 = R->cum[k6];
        const int ns = (int)R->cum[k6 + 1] - c0;
        if (ns > 8) { int nst_b = 0; cnt = mc_rt_lookup(X.rt, X.rt_mask, (uint32_t)bucket, qk, &nst_b); lb = nst_b - c0; }   // long group: the range table knows the answer
        else if (ns > 0) cnt = mc_group_match8(X.keys + start + c0, ns, qk, &lb);
    }
    return mc_en_append<McEnqWave, true>(X, item, cnt, c0 + lb, start, rbase, W, tasks, cap, counters, lane);
}

template <int MC_EN_WAVES>
__global__ void MC_EN_ATTR __launch_bounds__(64 * MC_EN_WAVES) k_enumerate_q(const McTables *__restrict__ T, McIndex X, const uint32_t *__restrict__ bitmap,
                                                                  const uint8_t *__restrict__ frames, int FP, int L, int64_t nreads, McSeedTask *tasks,
                                                                  uint32_t cap, uint32_t *counters, unsigned long long *stats, uint32_t read0, int prio
#ifdef MC_EN_FRONT_ONLY
                                                                  , uint32_t *wl_items, uint32_t wl_cap, uint32_t *wl_cursor
#endif
                                                                  )
{
    // read0: the launch searches reads read0 .. read0 + nreads - 1 of the range (frames points at the first of them) - a range is searched in
    // parts when the translation of the next part runs beside the search of this one (stage_a); prio: the waves' issue priority (s_setprio):
    // this kernel waits for scattered lines with two in five issue slots idle, the translation is bound by issue - above it, its waves get
    // their few instructions out at once and keep their requests in flight, and the translation takes the slots they leave
    if (prio == 3) __builtin_amdgcn_s_setprio(3); else if (prio == 2) __builtin_amdgcn_s_setprio(2); else if (prio == 1) __builtin_amdgcn_s_setprio(1);
#ifdef MC_EN_FRONT_ONLY
    __shared__ uint32_t fo_all[MC_EN_WAVES][16];
    uint32_t *fo_base = fo_all[threadIdx.x >> 6], *fo_used = fo_base + 8;
    if ((threadIdx.x & 63) < 8) { fo_base[threadIdx.x & 63] = 0; fo_used[threadIdx.x & 63] = 256u; }
#endif
    uint8_t *smem = mc_smem;
    uint8_t *grp = smem;                                                    // 32-byte group table
    const int lane = mc_lane(), wv = threadIdx.x >> 6;
    const int FPn = MC_EN_ROW(FP);
    const int ql0 = L / 3, ql1 = (L - 1) / 3, ql2 = (L - 2) / 3;           // frame lengths (frames f and f + 3 alike)
    const int cn0 = ql0 > 6 ? ql0 - 6 : 0, cn1 = ql1 > 6 ? ql1 - 6 : 0, cn2 = ql2 > 6 ? ql2 - 6 : 0;   // seed positions of the frames, and their running sums
    const int cum1 = cn0, cum2 = cum1 + cn1, cum3 = cum2 + cn2, cum4 = cum3 + cn0, cum5 = cum4 + cn1, cum6 = cum5 + cn2;
    const int nw32 = (cn0 + 31) >> 5;                                       // setter words per frame
    const uint32_t rcp_fpn = (65536u + (uint32_t)FPn - 1u) / (uint32_t)FPn;   // i / FPn = (i * rcp_fpn) >> 16 for the i < 6 * FPn in use
    uint8_t *wbase_lds = smem + 64 + (size_t)wv * MC_ENQ_WAVE_LDS(FP, L);
    McEnqWave *W = (McEnqWave *)wbase_lds;
    uint32_t *setter = (uint32_t *)(wbase_lds + sizeof(McEnqWave));
    uint8_t *fr = (uint8_t *)(setter + MC_ENQ_SETW(L));                    // the current read's rows of reduced-alphabet codes
    uint8_t *raw = fr + 6 * FPn;                                            // the next read's frames, on their way (global_load_lds)
    if (threadIdx.x < 32) grp[threadIdx.x] = T->grp[threadIdx.x];
    if (lane == 0) { W->blk_base = 0; W->blk_used = MC_EN_BLK; }
    __syncthreads();
    uint32_t ntasks = 0;                                                    // seed hits this lane appended
    uint32_t n_exact = 0, n_wild = 0, n_pairs = 0, n_probes = 0;         // what this wave asked its structures (wave-uniform)
    const unsigned long long lt = (1ull << lane) - 1;
    const int64_t nw = (int64_t)gridDim.x * MC_EN_WAVES;
    const int nraw = 6 * FP / 4;                                            // dwords of a read's frames (FP is a multiple of 4)
    // reads: chunks of MC_EN_CHUNK consecutive reads, the first by the wave's number, every further one from a counter (k_enumerate_t0)
    int64_t rnext = ((int64_t)blockIdx.x * MC_EN_WAVES + wv) * MC_EN_CHUNK, r = 0;
    int left = MC_EN_CHUNK;                                                 // reads of the current chunk not yet staged
    uint32_t pend = 0;                                                      // lane 0: the number of the chunk after this one
    bool next_same = false;                                                 // the read after the staged one belongs to the same chunk (and exists)
    int kpos = cum6;                                                        // next position of the staged read to decode (cum6: none left / no read staged)
    int qn = 0, en = 0, dn = 0, pn = 0;                                     // fills of q, eq, dq, pre
    uint32_t tag = 0, rbase = 0;                                            // the staged read's number inside its chunk; the chunk's first read
    uint32_t wm = 0, wdig = 0;                                              // groups of this lane's position that still have to enter eq; their own residues at the wildcard offsets
    unsigned long long wbase = 0;                                           // seed | key | position | frame | read of this lane's position
    uint32_t pm = 0;                                                        // surviving probes of this lane's expanded pair ...
    unsigned long long xi = 0;                                              // ... and the pair itself
    if (rnext < nreads) MC_EN_FETCH(rnext);
    for (;;) {
        const bool pmz = __ballot(pm != 0) == 0, wmz = __ballot(wm != 0) == 0;
        const bool drain = kpos >= cum6 && !next_same;                      // nothing more will enter pre / dq in this chunk
        const bool t3 = drain && pn == 0 && dn == 0 && wmz;                 // ... eq
        const bool t1 = t3 && en == 0 && pmz;                               // ... q
        if (qn >= 64 || (t1 && qn > 0)) {                                   // probes that passed the filters
            const int take = qn < 64 ? qn : 64;
            qn -= take;
            n_probes += (uint32_t)take;
            ntasks += mc_enq_process(X, (lane < take) ? W->q[qn + lane] : 0ull, lane < take, rbase, W, tasks, cap, counters, lane);
            mc_wave_sync();
            continue;
        }
        if (!pmz) {                                                         // queue the surviving probes: one per lane and turn, until q holds a full batch
            const int gc = (int)((xi >> 47) & 3), sd = (int)(xi & 0xFFFFF);
            const uint32_t xk = (uint32_t)((xi >> 20) & 0xFFFF);
            const int st = gc == 0 ? 10 : gc == 1 ? 1 : gc == 2 ? 100 : 0;
            const int dd = (int)((xi >> 53) & 15);
            const int s0 = sd - dd * st;                                    // the bucket with the substituted digit taken out (gc 3: the bucket itself)
            const unsigned long long keep = xi & 0x1E007FF000000000ull;     // position, frame, read
            for (;;) {
                const unsigned long long pmm = __ballot(pm != 0);
                if (pmm == 0 || qn >= 64) break;
                const int j = __builtin_ctz(pm | 0x400u);
                const int v = s0 + j * st;
                const uint32_t k2 = gc < 3 ? xk : ((xk & 0x0FFFu) | ((uint32_t)j << 12));
                if (pm) W->q[qn + __popcll(pmm & lt)] = keep | (unsigned long long)v | ((unsigned long long)k2 << 20) | ((unsigned long long)(1 + gc * 10 + j) << 47);
                qn += __popcll(pmm);
                pm &= pm - 1;
            }
            mc_wave_sync();
            continue;
        }
        if (en >= 64 || (t3 && en > 0)) {                                   // expand 64 (position, group) pairs into their ten probes: the pair filter
            const int take = en < 64 ? en : 64;
            en -= take;
            n_pairs += (uint32_t)take;
            const bool act = lane < take;
            xi = act ? W->eq[en + lane] : 0ull;
            const int gl = (int)((xi >> 47) & 3), sd = (int)(xi & 0xFFFFF);
            const uint32_t xk = (uint32_t)((xi >> 20) & 0xFFFF);
            const int d = (int)((xi >> 53) & 15);                           // the position's own residue at the wildcard offset
            const uint32_t hp = mc_pair_hash_d((uint32_t)sd, xk, gl, (uint32_t)d);
            const uint4 blk = ((const uint4 *)X.pair)[act ? mc_pair_block(hp) : 0u];   // (lanes without a pair read block 0)
            pm = act ? (mc_pair_test4(blk.x, blk.y, blk.z, blk.w, hp) & ~(1u << d) & 0x3FFu) : 0u;
            mc_wave_sync();
            continue;
        }
        if (!wmz) {                                                         // pending groups enter eq: one per lane and turn, until eq holds a full batch
            for (;;) {
                const unsigned long long wmm = __ballot(wm != 0);
                if (wmm == 0 || en >= 64) break;
                const int gl = __builtin_ctz(wm | 16u);
                if (wm) W->eq[en + __popcll(wmm & lt)] = wbase | ((unsigned long long)gl << 47) | ((unsigned long long)((wdig >> (4 * gl)) & 15u) << 53);
                en += __popcll(wmm);
                wm &= wm - 1;
            }
            mc_wave_sync();
            continue;
        }
        if (dn >= 64 || (drain && dn > 0)) {
            // 64 deferred positions: has the exact probe they depend on a range in the index?  Its 9-mer filter word and the position's
            // own wildcard line are asked together; bucket record and key group (or the range table) for those both let through.
            const int take = dn < 64 ? dn : 64;
            dn -= take;
            const bool here = lane < take;
            const unsigned long long e = W->dq[here ? dn + lane : 0];
            const uint32_t dkv = here ? W->dk[dn + lane] : 0xFFFFFFFFu;
            const bool ask = dkv != 0xFFFFFFFFu;
            const uint32_t seed = (uint32_t)(e & 0xFFFFF), qk = (uint32_t)(e >> 20) & 0xFFFFu;
            const uint32_t lo3 = seed % 1000u, d3 = lo3 / 100u, d4 = (lo3 / 10u) % 10u, d5 = lo3 % 10u;
            const uint32_t pb = dkv & 0xFFFFFu, pk = ((dkv >> 20) << 4) | 0xFu;   // the probe: bucket, key g6 g7 g8 F
            const unsigned long long am = __ballot(ask);
            n_exact += (uint32_t)__popcll(am); n_wild += (uint32_t)__popcll(am);
            const uint32_t hh = mc_filter_hash(pb, pk), fb9 = mc_filter_bits(hh);
            const uint32_t fw9 = X.filt[ask ? mc_filter9_word(hh) : 0u];
            const uint32_t ctx = mc_wild_ctx(seed, qk);
            const uint4 *ln = (const uint4 *)X.wild + (size_t)(ask ? mc_wild_line(ctx) : 0u) * 2;
            const uint4 q0 = ln[0], q1 = ln[1];
            const uint32_t wsum = mc_wild_sum(ctx, d3, d4, d5, qk >> 12);
            uint32_t wmt = 0;
            if (ask && (fw9 & fb9) == fb9)
                wmt = (mc_wild_test2(q0.x, q0.y, mc_wild_bits_s(wsum, d4, 0)) ? 1u : 0u) | (mc_wild_test2(q0.z, q0.w, mc_wild_bits_s(wsum, d5, 1)) ? 2u : 0u) |
                      (mc_wild_test2(q1.x, q1.y, mc_wild_bits_s(wsum, d3, 2)) ? 4u : 0u) | (mc_wild_test2(q1.z, q1.w, mc_wild_bits_s(wsum, qk >> 12, 3)) ? 8u : 0u);
            if (wmt) {                                                      // (few lanes: the filter passes one exact probe in ten)
                const McBucketRec *R = X.rec + pb;
                const int k6 = (int)(pk >> 12), c0 = R->cum[k6], ns = (int)R->cum[k6 + 1] - c0;
                int cnt = 0, lb = 0;
                if (ns > 8) cnt = mc_rt_lookup(X.rt, X.rt_mask, pb, pk, &lb);
                else if (ns > 0) cnt = mc_group_match8(X.keys + R->start + c0, ns, pk, &lb);
                if (cnt == 0) wmt = 0;
            }
            wm = wmt;
            wdig = d4 | (d5 << 4) | (d3 << 8) | ((qk >> 12) << 12);
            wbase = e;
            mc_wave_sync();
            continue;                                                       // (the groups enter eq in the state above)
        }
        if (pn >= 64 || (drain && pn > 0)) {
            // 64 kept positions, whatever reads they belong to: the exact 9-mer through its filter into q, the neighbourhood through the
            // wildcard filter (one 32-byte line answers for the four groups); both asked before either answer is looked at
            const int take = pn < 64 ? pn : 64;
            pn -= take;
            const bool here = lane < take;
            const unsigned long long pw = W->pre[here ? pn + lane : 0];
            const uint32_t tg = W->ptag[here ? pn + lane : 0];
            const uint32_t seed = (uint32_t)(pw & 0xFFFFF), qk = (uint32_t)(pw >> 20) & 0xFFFFu;
            const uint32_t d3 = (uint32_t)(pw >> 50) & 15u, d4 = (uint32_t)(pw >> 54) & 15u, d5 = (uint32_t)(pw >> 58) & 15u;   // bucket digits at offsets 3, 4, 5
            wdig = d4 | (d5 << 4) | (d3 << 8) | ((qk >> 12) << 12);         // the residue at the wildcard offset of groups 0..3
            wbase = (pw & 0x00007FFFFFFFFFFFull) | ((unsigned long long)tg << 57);   // seed | key | position | frame | read: a queue item without its phase
            const bool live0 = here && ((pw >> 47) & 1), live = here && ((pw >> 48) & 1);
            const unsigned long long m9 = __ballot(live0), mw = __ballot(live);
            n_exact += (uint32_t)__popcll(m9); n_wild += (uint32_t)__popcll(mw);
            const uint32_t qk0 = (qk & 0xFFF0u) | 0xFu;
            uint32_t fw9 = 0, fb9 = 0, wsum = 0;
            uint4 q0 = {0, 0, 0, 0}, q1 = {0, 0, 0, 0};
            if (m9) {                                                       // the exact 9-mer: its own Bloom filter, then straight into q
                const uint32_t hh = mc_filter_hash(seed, qk0);
                fb9 = mc_filter_bits(hh);
                fw9 = X.filt[live0 ? mc_filter9_word(hh) : 0u];
            }
#ifdef MC_EN_FRONT_ONLY
            // MEASUREMENT BUILD ONLY (the first pass of an XCD-sliced filter cascade priced): the wildcard asks are not made - each
            // becomes a 12-byte item (seed, key, position, frame, read) in one of eight lists by the top three bits of its filter line, in
            // blocks of 256 slots per wave and list (one global atomic per block).  The results of such a build are NOT the reference's.
            if (mw) {
                const uint32_t ctx = mc_wild_ctx(seed, qk);
                const uint32_t sl = mc_wild_line(ctx) >> (MC_WILD_LOG2L - 3);
                const uint32_t rd_ = rbase + tg;
#pragma unroll 1
                for (uint32_t s8 = 0; s8 < 8; s8++) {
                    const unsigned long long m8 = __ballot(live && sl == s8);
                    if (!m8) continue;
                    const uint32_t n8 = (uint32_t)__popcll(m8), r8 = (uint32_t)__popcll(m8 & lt);
                    uint32_t bb = fo_base[s8], bu = fo_used[s8];
                    mc_wave_sync();
                    if (bu + n8 > 256u) {
                        for (uint32_t i = bu + lane; i < 256u; i += 64) wl_items[((size_t)s8 * wl_cap + bb + i) * 3] = 0xFFFFFFFFu;
                        uint32_t nb = 0;
                        if (lane == 0) nb = atomicAdd(&wl_cursor[s8 * 32], 256u);
                        bb = (uint32_t)__builtin_amdgcn_readfirstlane((int)nb); bu = 0;
                        if (bb + 256u > wl_cap) { bb = 0; }                   // (measurement build: wrap instead of failing)
                    }
                    if (live && sl == s8) {
                        uint32_t *it = wl_items + ((size_t)s8 * wl_cap + bb + bu + r8) * 3;
                        it[0] = seed | ((uint32_t)((pw >> 36) & 0x7FFu) << 20); it[1] = qk; it[2] = rd_;
                    }
                    if (lane == 0) { fo_base[s8] = bb; fo_used[s8] = bu + n8; }
                    mc_wave_sync();
                }
            }
            const bool live_fo = false;
#define live live_fo
#else
            if (mw) {
                const uint32_t ctx = mc_wild_ctx(seed, qk);
                const uint4 *ln = (const uint4 *)X.wild + (size_t)(live ? mc_wild_line(ctx) : 0u) * 2;
                q0 = ln[0]; q1 = ln[1];
                wsum = mc_wild_sum(ctx, d3, d4, d5, qk >> 12);
            }
#endif
            const bool pr = live0 && (fw9 & fb9) == fb9;
            const unsigned long long prm = __ballot(pr);
            if (prm) {
                if (pr) W->q[qn + __popcll(prm & lt)] = wbase | (0xFull << 20);   // phase 0; key g6 g7 g8 F
                qn += __popcll(prm);
            }
            wm = 0;
            if (live) wm = (mc_wild_test2(q0.x, q0.y, mc_wild_bits_s(wsum, d4, 0)) ? 1u : 0u) | (mc_wild_test2(q0.z, q0.w, mc_wild_bits_s(wsum, d5, 1)) ? 2u : 0u) |
                           (mc_wild_test2(q1.x, q1.y, mc_wild_bits_s(wsum, d3, 2)) ? 4u : 0u) | (mc_wild_test2(q1.z, q1.w, mc_wild_bits_s(wsum, qk >> 12, 3)) ? 8u : 0u);
#ifdef MC_EN_FRONT_ONLY
#undef live
#endif
            for (;;) {   // the groups enter eq at once while it has room (else from the state above)
                const unsigned long long wmm = __ballot(wm != 0);
                if (wmm == 0 || en >= 64) break;
                const int gl = __builtin_ctz(wm | 16u);
                if (wm) W->eq[en + __popcll(wmm & lt)] = wbase | ((unsigned long long)gl << 47) | ((unsigned long long)((wdig >> (4 * gl)) & 15u) << 53);
                en += __popcll(wmm);
                wm &= wm - 1;
            }
            mc_wave_sync();
            continue;
        }
        if (kpos < cum6) {
            // The next slice of the staged read's positions (k_enumerate_t0's position pass): the 6-mer's bucket and the four key residues
            // (ten codes = 40 bits out of three aligned words of the row), whether the bucket holds anything (bitmap gather; those of
            // three sub-slices in flight together), and from that which probes the position makes.  Kept positions go to pre, deferred
            // ones to dq - completed below.
            const int k0 = kpos, dn0 = dn;
            kpos += MC_ENQ_SLICE;
            uint32_t sdv[3], gkv[3], bw[3], d3v[3], pfv[3];
            bool vd[3];
#pragma unroll
            for (int u = 0; u < 3; u++) {
                sdv[u] = 0; gkv[u] = 0; vd[u] = false; d3v[u] = 0; pfv[u] = 0;
                if (k0 + u * 64 >= cum6) continue;                             // (uniform)
                const int flat = k0 + u * 64 + lane;
                const int f = (flat >= cum1) + (flat >= cum2) + (flat >= cum3) + (flat >= cum4) + (flat >= cum5);
                const int pos = flat - (f == 0 ? 0 : f == 1 ? cum1 : f == 2 ? cum2 : f == 3 ? cum3 : f == 4 ? cum4 : cum5);
                const uint32_t *rw = (const uint32_t *)(fr + f * FPn) + (pos >> 3);   // (past the last position: some words of the wave's LDS, not used)
                const int o4 = (pos & 7) * 4;
                const uint32_t w0 = rw[0], w1 = rw[1], w2 = rw[2];
                unsigned long long v = (((unsigned long long)w1 << 32) | w0) >> o4;
                if (o4 == 28) v |= (unsigned long long)w2 << 36;
                const uint32_t six = (uint32_t)v & 0xFFFFFFu, y = six ^ 0xAAAAAAu;
                const bool bad = ((y - 0x111111u) & ~y & 0x888888u) != 0;      // one of the six codes is the invalid one
                const uint32_t seed = (six & 15u) * 100000u + ((six >> 4) & 15u) * 10000u + ((six >> 8) & 15u) * 1000u + ((six >> 12) & 15u) * 100u + ((six >> 16) & 15u) * 10u + (six >> 20);
                const uint32_t hi4 = (uint32_t)(v >> 24) & 0xFFFFu;            // g6 lowest
                const uint32_t gk = ((hi4 & 15u) << 12) | (((hi4 >> 4) & 15u) << 8) | (((hi4 >> 8) & 15u) << 4) | (hi4 >> 12);
                const bool ok = flat < cum6 && !bad;
                sdv[u] = ok ? seed : 0u; gkv[u] = gk; vd[u] = ok; d3v[u] = (six >> 12) & 0xFFFu; pfv[u] = (uint32_t)pos | ((uint32_t)f << 8);
            }
#pragma unroll
            for (int u = 0; u < 3; u++) bw[u] = bitmap[sdv[u] >> 5];
#pragma unroll
            for (int u = 0; u < 3; u++) {
                if (k0 + u * 64 >= cum6) break;
                const int pos = (int)(pfv[u] & 0xFF), f = (int)(pfv[u] >> 8), fm = f >= 3 ? f - 3 : f;
                const int rest = (fm == 0 ? cn0 : fm == 1 ? cn1 : cn2) - pos;   // residues behind the 6-mer
                const bool occ = (bw[u] >> (sdv[u] & 31)) & 1u;
                const uint32_t gk = gkv[u];
                const bool v6 = (gk >> 12) != MC_INVGRP, v7 = ((gk >> 8) & 15u) != MC_INVGRP, v8 = ((gk >> 4) & 15u) != MC_INVGRP, v9 = (gk & 15u) != MC_INVGRP;
                const bool live0 = vd[u] && occ && rest >= 3 && v6 && v7;        // exact 9-mer probe: it also defines `prev` for the positions behind it
                bool live = false, defer = false;                                // (see k_enumerate_count)
                if (vd[u] && rest >= 4) { if (occ) live = v6 && v7 && v9; else if (v8 && v9) { if (v6 && v7) live = true; else defer = true; } }
                if (live0) atomicOr(&setter[f * nw32 + (pos >> 5)], 1u << (pos & 31));
                const unsigned long long item = (unsigned long long)sdv[u] | ((unsigned long long)gk << 20) | ((unsigned long long)pfv[u] << 36);
                const bool keep = live0 || live;
                const unsigned long long km = __ballot(keep);
                if (keep) {
                    const int at = pn + __popcll(km & lt);
                    W->pre[at] = item | ((unsigned long long)(live0 ? 1u : 0u) << 47) | ((unsigned long long)(live ? 1u : 0u) << 48) | ((unsigned long long)d3v[u] << 50);
                    W->ptag[at] = (uint8_t)tag;
                }
                pn += __popcll(km);
                const unsigned long long dm = __ballot(defer);
                if (dm) {
                    if (defer) W->dq[dn + __popcll(dm & lt)] = item | ((unsigned long long)tag << 57);
                    dn += __popcll(dm);
                }
            }
            mc_wave_sync();
            // the deferred positions of this slice: the nearest earlier position of their frame that makes an exact probe (the setter flags
            // of everything in front of them are set by now), and that probe's bucket and key, read off the row
            for (int i = dn0 + lane; i < dn; i += 64) {
                const unsigned long long e = W->dq[i];
                const int pos = (int)((e >> 36) & 0xFF), fl = (int)((e >> 44) & 7);
                int w = pos >> 5;
                uint32_t m = setter[fl * nw32 + w] & ((1u << (pos & 31)) - 1);
                while (m == 0 && w > 0) { w--; m = setter[fl * nw32 + w]; }
                uint32_t dkv = 0xFFFFFFFFu;
                if (m) {
                    const int p2 = w * 32 + 31 - __builtin_clz(m);
                    const uint32_t *rw = (const uint32_t *)(fr + fl * FPn) + (p2 >> 3);
                    const int o4 = (p2 & 7) * 4;
                    const unsigned long long v = (((unsigned long long)rw[1] << 32) | rw[0]) >> o4;   // nine codes = 36 bits: they fit behind any o4 <= 28
                    const uint32_t six = (uint32_t)v & 0xFFFFFFu, k3 = (uint32_t)(v >> 24) & 0xFFFu;      // g6 lowest
                    const uint32_t seed = (six & 15u) * 100000u + ((six >> 4) & 15u) * 10000u + ((six >> 8) & 15u) * 1000u + ((six >> 12) & 15u) * 100u + ((six >> 16) & 15u) * 10u + (six >> 20);
                    dkv = seed | ((((k3 & 15u) << 8) | (((k3 >> 4) & 15u) << 4) | (k3 >> 8)) << 20);
                }
                W->dk[i] = dkv;
            }
            mc_wave_sync();
            continue;
        }
        // The staged read is decoded and fewer than a batch of everything is left (or, at the end of a chunk, nothing): the next read.
        if (rnext >= nreads) break;
        r = rnext;
        tag = (uint32_t)(r & (MC_EN_CHUNK - 1)); rbase = read0 + (uint32_t)r - tag;
        __builtin_amdgcn_s_waitcnt(0x0F70);                                  // vmcnt(0): this read's frames have arrived
        mc_wave_sync();
        {   // stage the six frames of this read as reduced-alphabet codes, two per byte (rows of FPn bytes, padded with the
            // invalid code: a seed's key residues past the frame end then read as invalid by themselves); clear the flags
            const uint8_t *src = raw;
            for (int i = lane; i < 6 * FPn; i += 64) {                         // byte i of the six rows
                const int f = (int)(((uint32_t)i * rcp_fpn) >> 16), b2 = 2 * (i - f * FPn);
                uint32_t g0 = MC_INVGRP, g1 = MC_INVGRP;
                if (b2 < FP) { const uint32_t two = *(const uint16_t *)(src + f * FP + b2); g0 = grp[two & 0xFF]; g1 = grp[two >> 8]; }   // (FP is a multiple of 4)
                fr[i] = (uint8_t)(g0 | (g1 << 4));
            }
            if (lane < 6 * nw32) setter[lane] = 0;
            mc_wave_sync();
        }
        // the read after this one: its frames start their way now (the next chunk is asked for one read ahead of need)
        left--;
        if (left == 1 && lane == 0) pend = atomicAdd(&counters[C_ENCHUNK], 1u);
        if (left > 0) { rnext = r + 1; next_same = rnext < nreads; }
        else { rnext = (nw + (int64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)pend)) * MC_EN_CHUNK; left = MC_EN_CHUNK; next_same = false; }
        if (rnext < nreads) MC_EN_FETCH(rnext);
        kpos = 0;
    }
    {   // close the wave's last block
        mc_wave_sync();
        const uint32_t bb = W->blk_base, bu = W->blk_used;
        for (uint32_t i = bu + lane; i < MC_EN_BLK; i += 64) tasks[bb + i].read = MC_TASK_NONE;
    }
    {
        unsigned long long c = ntasks;
        for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d);
        if (lane == 0) { atomicAdd(&stats[S_TASKS], c); atomicAdd(&stats[S_EXACT], (unsigned long long)n_exact); atomicAdd(&stats[S_WILD], (unsigned long long)n_wild); atomicAdd(&stats[S_PAIRS], (unsigned long long)n_pairs); atomicAdd(&stats[S_PROBES], (unsigned long long)n_probes); }
    }
}
