// k_translate_seg.h - stage A1: six-frame translation + SEG masking of the reads of a batch (BuildQHash@0x40b530, Seg::*).
#pragma once
#include "mc_hip_common.h"

// k_translate_seg: one thread per (read, frame); a workgroup is ONE wave and owns 10 consecutive reads (60 frames) - a workgroup
// of four waves (42 reads) waited for its slowest SEG: 5.75 against 5.27 ms per 1 M reads of 150 bp.
#define MC_TS_THREADS 64
#define MC_TS_WAVES (MC_TS_THREADS / 64)
#define MC_TS_READS (MC_TS_THREADS / 6)
// row pitch: an odd number of 32-bit words, so that the 64 lanes of a wave touching the same offset of their rows
// fall into different LDS banks (a pitch of 128 bytes put all of them into one)
#define MC_TS_NLNF(FP) ((FP) + 2 > 24 ? (FP) + 2 : 24)
#define MC_TS_STRIDE(FP) (((((FP) + 76 + 3) >> 2) | 1) << 2)
#define MC_TS_STAGE(L) ((((MC_TS_READS * (L)) > MC_TS_WAVES * 1488 ? (MC_TS_READS * (L)) : MC_TS_WAVES * 1488) + 15) & ~15)   // read staging, later one McSegWaveLds per wave

// ---- SEG for the 64 frames of a wave -----------------------------------------------------------------------------------
// mc_seg_mask_fx (mc_core.h) is the per-frame statement of the algorithm; this is the same algorithm arranged for a wave.
// The window scan and the bookkeeping of a frame stay with its lane (cheap, integer only).  What is expensive is the
// trimming of a low-complexity stretch of n residues - Seg::trim@0x439e20 evaluates Seg::getprob for every window of
// every length, n(n-1)/2 of them - and only one frame in five needs it, with very unequal n.  So the lanes stop when they
// reach a stretch, the windows of ALL stretches pending in the wave are numbered consecutively and dealt out to the 64
// lanes (each builds its window's composition from scratch, in registers for windows <= 15 residues), and the least
// probable window of every stretch (the first one in the reference's iteration order on a tie) is found with LDS
// atomics.  The double arithmetic of getprob is the reference's, operation by operation.
struct McSegWaveLds { unsigned long long best[64]; uint32_t pre[66]; uint32_t pre2[66]; uint32_t bq[64]; uint16_t off[64]; uint8_t n[64]; };   // 1,488 B per wave (it lies under the staged reads)
static_assert(sizeof(McHsp) == 48 && sizeof(McGapTask) % 4 == 0, "k_eval_seeds copies its staging buffers as 16- and 4-byte words");
static_assert(sizeof(McSegWaveLds) == 1488, "MC_TS_STAGE reserves 1488 bytes per wave");
#ifdef MC_EXP_TIMING
__device__ unsigned long long g_ts_acc[12], g_ts_cnt[12];
// (accumulated per wave in LDS and added to the global counters once at the end: an atomic per tick queues in front of the kernel's own loads and
// turns up as time of whichever phase touches global memory next)
#define MC_TS_TICK(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); if (lane == 0) { ts_acc_[tcat_] += now_ - tlast_; ts_acc_[12 + tcat_] += 1; } tlast_ = now_; tcat_ = (k); } while (0)
#define MC_TS_BEGIN(k) __shared__ unsigned long long ts_acc_[24]; if (lane < 24) ts_acc_[lane] = 0; mc_wave_sync(); unsigned long long tlast_ = __builtin_readcyclecounter(); int tcat_ = (k)
#define MC_TS_PARAMS , unsigned long long &tlast_, int &tcat_, unsigned long long *ts_acc_
#define MC_TS_ARGS , tlast_, tcat_, ts_acc_
#define MC_TS_END do { mc_wave_sync(); if (lane < 12) { atomicAdd(&g_ts_acc[lane], ts_acc_[lane]); atomicAdd(&g_ts_cnt[lane], ts_acc_[12 + lane]); } } while (0)
#else
#define MC_TS_TICK(k) do { } while (0)
#define MC_TS_BEGIN(k) do { } while (0)
#define MC_TS_PARAMS
#define MC_TS_ARGS
#define MC_TS_END do { } while (0)
#endif
#define MC_SEG_KEY_ONE 0xBFF0000000000000ull   // order-preserving key of 1.0 (the initial minprob of Seg::trim)

__device__ __forceinline__ unsigned long long mc_seg_key(double x)
{ // unsigned keys that order like the doubles
    const unsigned long long u = (unsigned long long)__double_as_longlong(x);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

__device__ __forceinline__ void mc_seg_wave(const double *lnf, const int32_t *fx, const uint64_t *__restrict__ segtab, uint8_t *prot, int n, bool act, const McSegWS ws, McSegWaveLds *WL,
                                         const uint8_t *lds0, int lane MC_TS_PARAMS)
{
    const int W = (n <= 11) ? 8 : 12;
    MC_TS_TICK(0);   // 0 flags 1 advance 2 numbering 3 class-0 rounds 4 class-1 rounds 5 reduction 6 owners 7 mask | the kernel: 8 staging 9 translation 10 write-out
    enum { POP = 0, SCAN = 1, WAIT = 2, DONE = 3 };
    int st = (act && W <= n) ? POP : DONE;
    // the window flags of the frame, once (mc_seg_mask_fx2 in mc_core.h is this function for one frame): every segment the
    // reference scans again reads its flags off them
    McBits192 Flo, Fhi, lo, nhi, mk;
    mc_bits_clear(Flo); mc_bits_clear(Fhi); mc_bits_clear(lo); mc_bits_clear(nhi); mc_bits_clear(mk);
    if (st != DONE) {
        mc_seg_window_flags_rg(fx, prot, n, W, Flo, Fhi);
        if (!(Flo.a | Flo.b | Flo.c)) st = DONE;
    }
    int sp = 1, base = 0, m = 0, i = 0, lowlim = 0, loi = 0, hii = 0;
    bool any = false;
    if (st != DONE) { ws.stk[0] = 0; ws.stk[1] = (int16_t)n; }
    const unsigned long long ltmask = (1ull << lane) - 1;
    for (;;) {
        MC_TS_TICK(1);
        // ---- every lane advances its own frame to the next stretch that needs trimming
        while (st == POP || st == SCAN) {
            if (st == POP) {
                if (sp == 0) { st = DONE; break; }
                sp--;
                base = ws.stk[2 * sp]; m = ws.stk[2 * sp + 1];
                if (W > m) continue;
                lo = mc_seg_flags_of(Flo, base, m, W);
                i = mc_bits_next(lo, 0);
                if (i >= m) continue;
                nhi = mc_bits_andnot(mc_bits_low(m), mc_seg_flags_of(Fhi, base, m, W));
                lowlim = 0; st = SCAN;
            }
            loi = mc_bits_prev(nhi, i) + 1; if (loi < lowlim) loi = lowlim;
            hii = mc_bits_next(nhi, i) - 1; if (hii > m - 1) hii = m - 1;
            st = WAIT;
        }
        const unsigned long long req = __ballot(st == WAIT);
        MC_TS_TICK(2);
        if (req == 0) break;
        // ---- number the windows of all pending stretches
        const int nreq = __popcll(req);
        const int myr = __popcll(req & ltmask);
        const int myn = hii - loi + 1;
        if (st == WAIT) {
            WL->off[myr] = (uint16_t)((prot + base + loi) - lds0);
            WL->n[myr] = (uint8_t)myn;
            WL->best[myr] = MC_SEG_KEY_ONE; WL->bq[myr] = 0xFFFFFFFFu;
        }
        mc_wave_sync();
        // One work item = up to R consecutive windows of one LENGTH of one stretch (Seg::trim: len = nn - j has j + 1 windows,
        // j = 0 .. nn - minlen - 1): the lane builds the composition of its first window and slides it (one residue out, one in),
        // keeping the first least probable window; the best of a stretch is then found with two LDS atomics per item.
        // A round takes as long as its longest item, and most rounds are far from full (a frame's stretches come one after the
        // other, so a wave goes through ~15 batches of a few stretches each): R = 1, 2, 4 or 8 is chosen per batch and class as
        // the smallest run for which the items still fit ONE round - the same windows, spread over more lanes.  Items of a
        // stretch are numbered by (j, run): j = R A + B has A + 1 runs, C(j) = R A (A + 1) / 2 + B (A + 1) items lie in front
        // of it.  Windows of up to 15 residues are evaluated in registers, longer ones on the lane's LDS row (several times
        // slower): the two kinds go in SEPARATE rounds - class 0: lengths <= 15 (j >= nn - 15), class 1: the others - so that
        // a round of register items does not wait for one LDS item.
#define MC_SEG_CJ(j, sh) (((((j) >> (sh)) * (((j) >> (sh)) + 1)) << (sh)) / 2 + ((j) & ((1 << (sh)) - 1)) * (((j) >> (sh)) + 1))
        int sh0 = 3, sh1 = 3;
        {   // lane r counts the items of stretch r for the four run lengths; prefix sums over the lanes give the numbering
            int K = 0, j0 = 0;
            if (lane < nreq) { const int nn = WL->n[lane], minlen = (nn - 100 > 1) ? nn - 100 : 1; K = nn - minlen; j0 = nn - 15 > 0 ? (nn - 15 < K ? nn - 15 : K) : 0; }
            uint32_t s0 = 0, s1 = 0;
#pragma unroll
            for (int sh = 2; sh >= 0; sh--) {                           // smallest run whose items fit one round (else 8)
                const uint32_t c1 = (uint32_t)MC_SEG_CJ(j0, sh), c0 = (uint32_t)MC_SEG_CJ(K, sh) - c1;
                const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)mc_wave_scan_add(c0), 63), t1 = (uint32_t)__builtin_amdgcn_readlane((int)mc_wave_scan_add(c1), 63);
                if (t0 <= 64) sh0 = sh;
                if (t1 <= 64) sh1 = sh;
            }
            {
                const uint32_t c1 = sh1 == 0 ? (uint32_t)MC_SEG_CJ(j0, 0) : sh1 == 1 ? (uint32_t)MC_SEG_CJ(j0, 1) : sh1 == 2 ? (uint32_t)MC_SEG_CJ(j0, 2) : (uint32_t)MC_SEG_CJ(j0, 3);
                const uint32_t cj0 = sh0 == 0 ? (uint32_t)MC_SEG_CJ(j0, 0) : sh0 == 1 ? (uint32_t)MC_SEG_CJ(j0, 1) : sh0 == 2 ? (uint32_t)MC_SEG_CJ(j0, 2) : (uint32_t)MC_SEG_CJ(j0, 3);
                const uint32_t ck = sh0 == 0 ? (uint32_t)MC_SEG_CJ(K, 0) : sh0 == 1 ? (uint32_t)MC_SEG_CJ(K, 1) : sh0 == 2 ? (uint32_t)MC_SEG_CJ(K, 2) : (uint32_t)MC_SEG_CJ(K, 3);
                s0 = mc_wave_scan_add(ck - cj0); s1 = mc_wave_scan_add(c1);    // lengths <= 15; lengths > 15 (j < j0)
            }
            if (lane == 0) { WL->pre[0] = 0; WL->pre2[0] = 0; }
            WL->pre[lane + 1] = s0; WL->pre2[lane + 1] = s1;            // (entries past nreq repeat the total)
        }
        mc_wave_sync();
        for (int cls = 0; cls < 2; cls++) {
        const uint32_t *pre = cls ? WL->pre2 : WL->pre;
        const uint32_t total = pre[nreq];
        const int sh = cls ? sh1 : sh0, R = 1 << sh;
        for (uint32_t p0 = 0; p0 < total; p0 += 64) {
            MC_TS_TICK(3 + cls);
            const uint32_t p = p0 + (uint32_t)lane;
            const bool ok = p < total;
            int r = 0;
            for (int stp = 32; stp > 0; stp >>= 1) { const int k = r + stp; if (k < nreq && p >= pre[k]) r = k; }   // the stretch item p belongs to: last r with pre[r] <= p
            if (!ok) r = 0;
            const int nn = WL->n[r];
            int x = (int)(p - pre[r]);
            if (cls == 0) { const int minlen = (nn - 100 > 1) ? nn - 100 : 1, K = nn - minlen, j0 = nn - 15 > 0 ? (nn - 15 < K ? nn - 15 : K) : 0; x += MC_SEG_CJ(j0, sh); }
            // x = R A (A + 1) / 2 + B (A + 1) + run: the largest A with R A (A + 1) / 2 <= x
            int A = (int)((sqrtf(1.0f + 8.0f * (float)x / (float)R) - 1.0f) * 0.5f);
            while ((((A + 1) * (A + 2)) << sh) / 2 <= x) A++;
            while (((A * (A + 1)) << sh) / 2 > x) A--;
            const int rem = x - ((A * (A + 1)) << sh) / 2, B = rem / (A + 1), run = rem - B * (A + 1);
            const int j = (A << sh) + B, wfirst = run << sh, wlast = (wfirst + R - 1 < j) ? wfirst + R - 1 : j;
            const uint8_t *s = lds0 + WL->off[r];
            const int len = nn - j;
            const uint32_t qbase = (uint32_t)(j * (j + 1) / 2);          // number of window 0 of this length in Seg::trim's order
            unsigned long long key = MC_SEG_KEY_ONE;                     // (minprob starts at 1.0: only a smaller probability counts)
            uint32_t kq = 0xFFFFFFFFu;
            if (ok) {
                if (cls == 0) {
                    McRhState rg; rg.clo = 0; rg.chi = 0; rg.hist = 0;
                    for (int k = 0; k < len; k++) mc_rh_add(rg, s[wfirst + k]);
                    // the state vectors of the run first, then their table reads (in flight together), then the comparison in window order
                    const int cnt = wlast - wfirst + 1;
                    uint64_t svs[8];
#pragma unroll
                    for (int t = 0; t < 8; t++) {
                        svs[t] = rg.hist | ((uint64_t)len << 60);
                        if (t + 1 < cnt) { mc_rh_remove(rg, s[wfirst + t]); mc_rh_add(rg, s[wfirst + t + len]); }
                    }
                    // a pair lies in one of two slots (mc_segtab_slots): both are fetched, four windows' worth in flight at a time
#pragma unroll
                    for (int t0 = 0; t0 < 8; t0 += 4) {
                        if (t0 && R <= 4) break;                             // (R is the same for the whole wave)
                        ulonglong2 ea[4], eb[4];
#pragma unroll
                        for (int t = 0; t < 4; t++) {
                            uint32_t h1, h2;
                            mc_segtab_slots(svs[t0 + t], h1, h2);
                            const bool in = t0 + t < cnt;
                            ea[t] = ((const ulonglong2 *)segtab)[in ? h1 : 0u]; eb[t] = ((const ulonglong2 *)segtab)[in ? h2 : 0u];
                        }
#pragma unroll
                        for (int t = 0; t < 4; t++)
                            if (t0 + t < cnt) {
                                const unsigned long long pk = ea[t].x == svs[t0 + t] ? ea[t].y : eb[t].y;
                                if (pk < key) { key = pk; kq = qbase + (uint32_t)(wfirst + t0 + t); }
                            }
                    }
                } else {
                    mc_seg_comp_rg(s + wfirst, len, ws.comp);
                    mc_seg_state(ws.comp, ws.sv);
                    for (int w0 = wfirst;; w0++) {
                        const unsigned long long k2 = mc_seg_key(mc_seg_getprob(lnf, ws.sv, len));
                        if (k2 < key) { key = k2; kq = qbase + (uint32_t)w0; }
                        if (w0 == wlast) break;
                        mc_seg_shift(ws.comp, ws.sv, s[w0], s[w0 + len]);
                    }
                }
            }
            MC_TS_TICK(5);
            const bool cand = ok && key < MC_SEG_KEY_ONE;
            const unsigned long long old = WL->best[r];
            mc_wave_sync();
            if (cand) atomicMin(&WL->best[r], key);
            mc_wave_sync();
            const unsigned long long nb = WL->best[r];
            if (ok && nb != old) WL->bq[r] = 0xFFFFFFFFu;             // a smaller probability appeared in this round: forget the old window
            mc_wave_sync();
            if (cand && key == nb) atomicMin(&WL->bq[r], kq);
            mc_wave_sync();
        }
        }
#undef MC_SEG_CJ
        // ---- the owners take their results and go on
        MC_TS_TICK(6);
        if (st == WAIT) {
            const uint32_t q = WL->bq[myr];
            int lend = 0, rend = myn - 1;
            if (q != 0xFFFFFFFFu) {
                int j = (int)((sqrtf((float)(8u * q + 1u)) - 1.0f) * 0.5f);
                while ((uint32_t)((j + 1) * (j + 2) / 2) <= q) j++;
                while ((uint32_t)(j * (j + 1) / 2) > q) j--;
                const int len = myn - j, w0 = (int)q - j * (j + 1) / 2;
                lend = w0; rend = len + w0 - 1;
            }
            const int leftend = loi + lend, rightend = hii - (myn - rend - 1);
            if (i < leftend) {
                const int l2 = loi, r2 = leftend - 1;
                if (sp < 8) { ws.stk[2 * sp] = (int16_t)(base + l2); ws.stk[2 * sp + 1] = (int16_t)(r2 - l2 + 1); sp++; }
            }
            mk = mc_bits_or(mk, mc_bits_range(base + leftend, base + rightend));
            any = true;
            lowlim = ((hii < rightend) ? hii : rightend) + 1;
            i = mc_bits_next(lo, lowlim);
            st = i < m ? SCAN : POP;
        }
        mc_wave_sync();
    }
    MC_TS_TICK(7);
    if (any) for (int k = 0; k < n; k++) if (mc_bits_test(mk, k)) prot[k] = MC_INV;
    MC_TS_TICK(10);
}

// mc_translate_frame for a lane of k_translate_seg.  The plain form reads three bases, walks two compare chains per base, looks the
// codon up in the tables in global memory and stores one byte - and as the bases and the frame are both bytes in LDS, every store
// orders the loads behind it: one codon at a time at the latency of a global load, half of the kernel's time (cycle counters).
// Here: the codon table lies in LDS (cod, 64 bytes), the bases of 8 codons are read together, indices come from mc_nt_code
// (shifts and masks), the 8 residues leave as two words.  prot is 4-byte aligned.
__device__ __forceinline__ int mc_translate_frame_lds(const uint8_t *cod, const uint8_t *read, int len, int frame, uint8_t *prot)
{
    const int o = frame % 3;
    int n = (len - o) / 3;
    if (n < 0) n = 0;
    const bool rc = frame >= 3;
    const uint32_t set = rc ? MC_NT_RC_SET : MC_NT_FWD_SET, perm = rc ? MC_NT_RC_PERM : MC_NT_FWD_PERM;
    const uint8_t *p = read + (rc ? len - 1 - o : o);                // base k of the frame: p[k] forward, p[-k] on the reverse strand
    const int s = rc ? -1 : 1;
    int i = 0;
    for (; i + 8 <= n; i += 8) {
        uint32_t b[24], w[2] = {0, 0};
#pragma unroll
        for (int k = 0; k < 24; k++) b[k] = p[s * (3 * i + k)];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int a0 = mc_nt_code(b[3 * k], set, perm), a1 = mc_nt_code(b[3 * k + 1], set, perm), a2 = mc_nt_code(b[3 * k + 2], set, perm);
            const uint32_t aa = cod[(16 * a0 + 4 * a1 + a2) & 63];
            w[k >> 2] |= ((a0 | a1 | a2) < 0 ? (uint32_t)MC_INV : aa) << (8 * (k & 3));
        }
        *(uint32_t *)(prot + i) = w[0];
        *(uint32_t *)(prot + i + 4) = w[1];
    }
    for (; i < n; i++) {
        const int a0 = mc_nt_code(p[s * (3 * i)], set, perm), a1 = mc_nt_code(p[s * (3 * i + 1)], set, perm), a2 = mc_nt_code(p[s * (3 * i + 2)], set, perm);
        const uint8_t aa = cod[(16 * a0 + 4 * a1 + a2) & 63];
        prot[i] = (a0 | a1 | a2) < 0 ? (uint8_t)MC_INV : aa;
    }
    return n;
}

// One thread per (read, frame).  The workgroup's reads are staged into LDS with coalesced loads, every thread translates its
// frame into its own LDS row, the wave runs SEG on its frames (mc_seg_wave) and writes them back with coalesced stores.
// LDS per workgroup: max(10 L, 1,488) + ln n! + 64 (FP + 76) bytes (~10 KB at 150 bp; the staging area is reused by the SEG
// queues) - the registers (127) allow 16 waves per CU, the LDS holds 15.
template <bool STAGED>                                           // STAGED: the block's reads go through LDS (coalesced); otherwise each thread
__global__ void __attribute__((amdgpu_waves_per_eu(4, 4))) __launch_bounds__(MC_TS_THREADS) k_translate_seg(const McTables *__restrict__ T, const uint8_t *__restrict__ reads, int L,   // walks its read in global memory and the LDS it saves buys a workgroup per CU (long reads)
                                                       int64_t nreads, uint8_t *__restrict__ frames, int FP, const uint64_t *__restrict__ segtab)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * MC_TS_READS;
    const int nr = (int)((nreads - r0) < MC_TS_READS ? (nreads - r0) : MC_TS_READS);
    const int rbytes = nr * L;
    const int stride = MC_TS_STRIDE(FP);                         // per-thread LDS row: prot[FP] comp[20] sv[24] stk[32]
    __shared__ int32_t fxs[64];                                  // fixed-point entropy tables (mc_seg_mask_fx)
#ifdef MC_EXP_TIMING
    const int lane = mc_lane();
#endif
    MC_TS_BEGIN(8);
    uint8_t *sreads = smem;
    const int nlnf = MC_TS_NLNF(FP);
    double *lnf = (double *)(smem + (STAGED ? MC_TS_STAGE(L) : MC_TS_STAGE(0)));   // ln n! for n <= max(frame length, 20): all the trimming asks for
    uint8_t *rows = (uint8_t *)(lnf + nlnf);
    __shared__ __attribute__((aligned(4))) uint8_t cod[64];      // the codon table
    if (tid < 64) fxs[tid] = T->seg_dout[tid];                   // seg_dout, seg_din, seg_tlo, seg_thi are contiguous
    if (tid < 16) ((uint32_t *)cod)[tid] = ((const uint32_t *)T->codon)[tid];
    for (int i = tid; i < nlnf; i += MC_TS_THREADS) lnf[i] = T->lnfac[i];
    if (STAGED) {   // coalesced staging of this block's reads: 4 bytes per lane where the slice allows it (it starts at r0*L: any alignment)
        const uint8_t *src = reads + r0 * L;
        const int head = (int)((4 - ((uintptr_t)src & 3)) & 3), nhead = head < rbytes ? head : rbytes;
        if (tid < nhead) sreads[tid] = src[tid];
        const int nw = (rbytes - nhead) >> 2;
        if (nhead == 0) for (int i = tid; i < nw; i += MC_TS_THREADS) ((uint32_t *)sreads)[i] = ((const uint32_t *)src)[i];
        else for (int i = tid; i < nw; i += MC_TS_THREADS) { const uint32_t w = ((const uint32_t *)(src + nhead))[i]; uint8_t *d = sreads + nhead + 4 * i; d[0] = (uint8_t)w; d[1] = (uint8_t)(w >> 8); d[2] = (uint8_t)(w >> 16); d[3] = (uint8_t)(w >> 24); }
        for (int i = nhead + 4 * nw + tid; i < rbytes; i += MC_TS_THREADS) sreads[i] = src[i];
    }
    __syncthreads();
    MC_TS_TICK(9);
    const int lr = tid / 6, f = tid - lr * 6;
    uint8_t *prot = rows + (size_t)tid * stride;
    int n = 0;
    if (lr < nr) n = mc_translate_frame_lds(cod, STAGED ? sreads + lr * L : reads + (r0 + lr) * L, L, f, prot);
    __syncthreads();                                             // the staged reads are dead: their space becomes the SEG queues
    {
        McSegWS ws; ws.comp = prot + FP; ws.sv = prot + FP + 20; ws.stk = (int16_t *)(prot + FP + 44);
        mc_seg_wave(lnf, fxs, segtab, prot, n, lr < nr, ws, (McSegWaveLds *)smem + (tid >> 6), rows, mc_lane() MC_TS_ARGS);   // (stretch offsets are kept relative to the rows: 256 x 252 bytes at most, 16 bits)
        if (lr < nr) for (int i = n; i < FP; i++) prot[i] = MC_INV;
    }
    __syncthreads();
    {   // frames of the block are contiguous in global memory: nr*6 rows of FP bytes
        uint32_t *dst = (uint32_t *)(frames + r0 * 6 * FP);          // (FP and the LDS row pitch are multiples of 4: a word never straddles two rows)
        const int total = nr * 6 * FP / 4, fpw = FP / 4;
        for (int i = tid; i < total; i += MC_TS_THREADS) { const int row = i / fpw, col = i - row * fpw; dst[i] = *(const uint32_t *)(rows + (size_t)row * stride + 4 * col); }
    }
    MC_TS_TICK(0);
    MC_TS_END;
}
