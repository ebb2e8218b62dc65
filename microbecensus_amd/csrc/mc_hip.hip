// mc_hip.hip - HIP kernels (gfx950) and the C ABI of libmcensus_hip.so.
//
// Pipeline for one batch of reads resident in HBM (all stages on the handle's stream):
//   k_translate_seg   1 thread / (read, frame)   6-frame translation + SEG masking      -> frames
//   k_enumerate       1 thread / (read, frame)   reduced-alphabet seeds, bucket probes  -> seed tasks
//   k_eval_seeds      1 thread / seed task       seed gate, growth, ungapped X-drop     -> HSPs | gap tasks
//   k_gapped          1 thread / gap task        trace-free affine X-drop, both flanks  -> HSPs
//   radix sort        (read, subject, chrono)    rocPRIM device sort of the HSP keys
//   k_finish          1 thread / read with HSPs  linking, ranking, cap, classification   -> rows, best hits
// The per-thread algorithms live in mc_core.h / mc_finish.h; see include/mcensus.h for what each entry
// point replaces in the reference.
#include <cstddef>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mcensus.h"
#include "mc_finish.h"
#include "mc_index.h"

static_assert(sizeof(McRow) % 8 == 0 && sizeof(McRow) == sizeof(mc_row) && offsetof(McRow, ident) == offsetof(mc_row, ident) && offsetof(McRow, loge) == offsetof(mc_row, loge) &&
                  offsetof(McRow, score) == offsetof(mc_row, score) && offsetof(McRow, frame) == offsetof(mc_row, nmatch),
              "the device row is handed out as the ABI row");

static thread_local std::string g_err;
// MC_OPEN_TIMING in the environment: where the time of opening an engine and of its first run goes (stderr; development aid)
static double mc_now() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
static bool mc_open_timing() { static const bool on = getenv("MC_OPEN_TIMING") != nullptr; return on; }
#define MC_OT(label, t0) do { if (mc_open_timing()) { const double now_ = mc_now(); fprintf(stderr, "open-timing %-28s %8.1f ms\n", label, (now_ - (t0)) * 1e3); (t0) = now_; } } while (0)
extern "C" const char *mc_last_error(void) { return g_err.c_str(); }

#define HIPCK(call)                                                                                         \
    do {                                                                                                    \
        hipError_t e_ = (call);                                                                             \
        if (e_ != hipSuccess) {                                                                             \
            g_err = std::string(#call) + ": " + hipGetErrorString(e_);                                       \
            return -1;                                                                                      \
        }                                                                                                   \
    } while (0)

enum { C_TASKS = 0, C_GAPS, C_HSPS, C_HEADS, C_ROWS, C_OVERFLOW, C_SEGS, C_BEST, C_RETRY, C_HEAVY, C_HEAVY2, C_ITEMS, C_RETRY2, C_HEAVY3, C_LIGHT0, C_LIGHT1, C_LIGHT2, C_LIGHT3, C_HSPS2, C_HPAD, C_GPAD, C_ORDER, C_ORDER2, C_ORDER3, C_OTAKE, C_OTAKE2, C_OTAKE3, C_N = 28 };
enum { S_LOOKUPS = 0, S_KEYPROBES, S_TASKS, S_EXACT = 16, S_WILD, S_PAIRS, S_PROBES, S_N = 20 };   // 64-bit algorithmic-traffic counters of k_enumerate; slots 4..: cycle counters of the MC_EXP_TIMING build

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
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

__device__ __forceinline__ int mc_lane() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
// orders the wave's own LDS traffic for the compiler; the hardware executes one wave's LDS instructions in order
// inclusive prefix sum over the 64 lanes in six DPP additions: shifts inside the rows of 16, then the row totals carried across
__device__ __forceinline__ uint32_t mc_wave_scan_add(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);    // row_shr:1 (lanes shifted in from outside a row read 0)
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);    // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);    // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);    // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);   // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);   // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ void mc_wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

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

// one atomic per wave: the lanes with want == true receive consecutive slots of a global counter
__device__ __forceinline__ uint32_t mc_wave_alloc(uint32_t *counter, bool want)
{
    const unsigned long long m = __ballot(want);
    if (m == 0) return 0;
    const int lane = mc_lane(), leader = __builtin_ctzll(m);
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
    return base + (uint32_t)__popcll(m & ((1ull << lane) - 1));
}

// one atomic per 256-thread workgroup (every thread of the block must call it): a device-scope atomic on ONE counter runs at
// the memory side at ~125 M/s, so even one per wave is too many for kernels of millions of threads
__device__ __forceinline__ uint32_t mc_block_alloc(uint32_t *counter, bool want)
{
    __shared__ uint32_t wcnt[4], wbase[4];
    const unsigned long long m = __ballot(want);
    const int lane = mc_lane(), wv = threadIdx.x >> 6;
    if (lane == 0) wcnt[wv] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t c0 = wcnt[0], c1 = wcnt[1], c2 = wcnt[2], c3 = wcnt[3], tot = c0 + c1 + c2 + c3;
        const uint32_t b = tot ? atomicAdd(counter, tot) : 0u;
        wbase[0] = b; wbase[1] = b + c0; wbase[2] = b + c0 + c1; wbase[3] = b + c0 + c1 + c2;
    }
    __syncthreads();
    const uint32_t r = wbase[wv] + (uint32_t)__popcll(m & ((1ull << lane) - 1));
    __syncthreads();                                             // (the arrays are reused by the next call)
    return r;
}

// copies the hot members of the tables into LDS (block-wide; callers __syncthreads() afterwards)
__device__ __forceinline__ void mc_load_hot(McHot *H, const McTables *T)
{
    for (int i = threadIdx.x; i < 32 * 32 / 4; i += blockDim.x) ((uint32_t *)H->sub)[i] = ((const uint32_t *)T->sub)[i];
    if (threadIdx.x < 32) H->grp[threadIdx.x] = T->grp[threadIdx.x];
    if (threadIdx.x == 0) { H->xdrop_ungapped = T->xdrop_ungapped; H->xdrop_gapped = T->xdrop_gapped; H->gap_trigger = T->gap_trigger; }
}

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
            t.read = read; t.chrono = MC_CHRONO(frame, pos, phase, nst + i); t.posting = X->post[b0 + nst + i];
            t.seedlen_nkey = MC_TASK_W3(X->off[t.posting >> 11] + (t.posting & 0x7ff), seedlen, nkey);
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
// k_enumerate_t0: position-parallel seed probing for databases whose .info frequency threshold is 0 (the marker DB).
//
// With threshold 0 the seed-length carry of Searching@0x415050 collapses (mc_enumerate_seeds documents the general
// rule): a position whose own bucket is non-empty always uses a 9-mer (or is skipped), and only positions with an
// EMPTY bucket look at `prev` - to decide from where the 10-mer validity check of the neighbourhood starts.  `prev`
// is 9 if the nearest earlier non-skipped position with a non-empty bucket found a matching 9-mer range, else 6.
// So a read is handled in two parallel passes: (0) the exact 9-mer probes and the 36 neighbourhood probes of every
// position whose neighbourhood does not depend on `prev`; (1) the few positions that do.
//
// One wave per read, 24 waves per CU.  The kernel is bound by instruction issue (VALU + SALU), not by memory: every stage
// is arranged so that all 64 lanes work - positions, (position, wildcard offset) pairs and probes are compacted through
// per-wave LDS queues - and so that a stage costs few instructions per item (filters that answer in one read, packed
// codes, prefix sums by DPP).  Seed hits are appended to slots from a prefix sum; one global atomic per 2048 slots.
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
    unsigned long long hq[MC_EN_QCAP];      // probes whose first-residue group is longer than 8 keys (binary search): counting form only - last member, not allocated otherwise
};
#define MC_EN_WAVE_BYTES(COUNT) ((COUNT) ? sizeof(McEnWave) : offsetof(McEnWave, hq))

extern __shared__ __attribute__((aligned(16))) uint8_t mc_smem[];   // dynamic LDS of the kernels that use it

// item: bucket(20) | qk(16)<<20 | pos(8)<<36 | frame(3)<<44 | phase(6)<<47
// Appends the seed hits of one batch of probes (lane: cnt postings starting at posting index nst of its bucket).
__device__ __forceinline__ uint32_t mc_en_append(const McIndex &X, unsigned long long item, int cnt, int nst, uint32_t start, uint32_t read, McEnWave *W,
                                                 McSeedTask *tasks, uint32_t cap, uint32_t *counters, int lane)
{
    unsigned long long m = __ballot(cnt > 0);
    if (m == 0) return 0;
    const int pos = (int)((item >> 36) & 0xFF), frame = (int)((item >> 44) & 7), phase = (int)((item >> 47) & 63);
    if (phase == 0 && cnt > 0) atomicOr(&W->hit[frame][pos >> 5], 1u << (pos & 31));
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
        uint32_t pst[MC_EN_SHORT], ofs[MC_EN_SHORT];   // the postings and the subjects' offsets first, then the stores: a store between two loads orders them (the pointers may alias)
#pragma unroll
        for (int i = 0; i < MC_EN_SHORT; i++) pst[i] = X.post[start + (uint32_t)nst + (uint32_t)(i < cnt ? i : 0)];
#pragma unroll
        for (int i = 0; i < MC_EN_SHORT; i++) ofs[i] = X.off[pst[i] >> 11];
#pragma unroll
        for (int i = 0; i < MC_EN_SHORT; i++)
            if (i < cnt) {
                McSeedTask t;
                t.read = read; t.chrono = MC_CHRONO(frame, pos, phase, (uint32_t)nst + (uint32_t)i); t.posting = pst[i];
                t.seedlen_nkey = sn | (ofs[i] + (pst[i] & 0x7ffu));
                tasks[base + excl + (uint32_t)i] = t;
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
                uint32_t pst[2], slot[2], chr[2], snk[2];
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
                    pst[u] = X.post[in[u] ? ofrom + i : 0u];
                    slot[u] = base + oex + i; chr[u] = MC_CHRONO(f2, p2, ph2, onst + i); snk[u] = ph2 == 0 ? MC_TASK_W3(0, 9, 3) : MC_TASK_W3(0, 10, 4);
                }
#pragma unroll
                for (int u = 0; u < 2; u++) snk[u] |= X.off[pst[u] >> 11] + (pst[u] & 0x7ffu);
#pragma unroll
                for (int u = 0; u < 2; u++)
                    if (in[u]) {
                        McSeedTask t;
                        t.read = read; t.chrono = chr[u]; t.posting = pst[u]; t.seedlen_nkey = snk[u];
                        tasks[slot[u]] = t;
                    }
            }
        }
    }
    return (uint32_t)cnt;
}

// One batch of (up to 64) probes.  Returns per lane: key reads of the reference (bits 32..), seed hits (bits 8..31);
// bits 0..7 (uniform): the new fill of the heavy queue.
template <bool COUNT>
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
        if (ns > 0 && !heavy) cnt = COUNT ? mc_group_range8(X.keys + start + c0, ns, qk, &lb) : mc_group_match8(X.keys + start + c0, ns, qk, &lb);   // (the counting form wants the lower bound of an empty range too)
        if (!COUNT && heavy) {                                    // long group: the range table knows the answer (no binary search, no second queue)
            int nst_b = 0;
            cnt = mc_rt_lookup(X.rt, X.rt_mask, (uint32_t)bucket, qk, &nst_b);
            lb = nst_b - c0;
            heavy = false;
        }
        if (COUNT && !heavy) { const int n = R->cum[11]; kp = mc_bsearch_reads(n, c0 + lb) + (cnt > 0 ? mc_bsearch_reads(n, c0 + lb + cnt) : 0u); }
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
    const uint32_t nt = mc_en_append(X, item, cnt, c0 + lb, start, read, W, tasks, cap, counters, lane);
    return ((unsigned long long)kp << 32) | ((unsigned long long)nt << 8) | (unsigned long long)hn;
}

// One batch of probes whose group needs the binary searches.
template <bool COUNT>
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
        if (COUNT) { const int n = R->cum[11]; kp = mc_bsearch_reads(n, c0 + lb) + (cnt > 0 ? mc_bsearch_reads(n, c0 + lb + cnt) : 0u); }
    }
    const uint32_t nt = mc_en_append(X, item, cnt, c0 + lb, start, read, W, tasks, cap, counters, lane);
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
template <int MC_EN_WAVES, bool COUNT>
__global__ void MC_EN_ATTR __launch_bounds__(64 * MC_EN_WAVES) k_enumerate_t0(const McTables *__restrict__ T, McIndex X, const uint32_t *__restrict__ bitmap,
                                                                   const uint8_t *__restrict__ frames, int FP, int L, int64_t nreads, McSeedTask *tasks,
                                                                   uint32_t cap, uint32_t *counters, unsigned long long *stats)
{
    uint8_t *smem = mc_smem;
    uint8_t *grp = smem;                                                    // 32-byte group table
    const int lane = mc_lane(), wv = threadIdx.x >> 6;
    McEnWave *W = (McEnWave *)(smem + 64 + (size_t)wv * MC_EN_WAVE_BYTES(COUNT));
    uint8_t *fr_all = smem + 64 + (size_t)MC_EN_WAVES * MC_EN_WAVE_BYTES(COUNT);
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
    if ((int64_t)blockIdx.x * MC_EN_WAVES + wv < nreads) MC_EN_FETCH((int64_t)blockIdx.x * MC_EN_WAVES + wv);
    for (int64_t r = (int64_t)blockIdx.x * MC_EN_WAVES + wv; r < nreads; r += nw) {
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
        if (r + nw < nreads) MC_EN_FETCH(r + nw);
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
                if (COUNT && vd[u]) sc.lookups += live0 ? 2 : 1;               // bucket-size probe of the exact seed, and its key-range probe
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
        // the nearest earlier exact probe of the frame found a range; those wait in the list dq - with the wildcard
        // filter's answer, asked in pass 0 - until pass 0 has drained its queues, and are generated in pass 1.
        // With the counters off, filters decide what is searched:
        //   exact 9-mer  -> 9-mer Bloom filter -> queue q
        //   10-mers      -> wildcard filter (one 32-byte line per position answers for its four groups) -> queue eq of
        //                   (position, group) pairs -> 64 pairs at a time: pair filter (one 16-byte block answers for the
        //                   ten residues of the pair) -> queue q
        //   q            -> bucket records: group scan, or the range table for long groups -> seed hits
        // (the counting form searches every probe; its long groups go through queue hq to the binary searches).
        // Every stage runs with full waves; the generator is a state machine so that each stage exists once in the kernel.
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
                    const unsigned long long rh = mc_en_heavy<COUNT>(X, (lane < take) ? W->hq[hn + lane] : 0ull, lane < take, (uint32_t)r, W, tasks, cap, counters, lane);
                    sc.keyprobes += (uint32_t)(rh >> 32); sc.tasks += (uint32_t)(rh >> 8) & 0xFFFFFFu;
                    mc_wave_sync();
                    continue;
                }
                if (qn >= 64 || (tail && qn > 0)) {                      // probes that passed the filters
                    MC_TICK(2);
                    const int take = qn < 64 ? qn : 64;
                    qn -= take;
                    n_probes += (uint32_t)take;
                    const unsigned long long ret = mc_en_process<COUNT>(X, (lane < take) ? W->q[qn + lane] : 0ull, lane < take, (uint32_t)r, W, hn, tasks, cap, counters, lane
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
                    const uint32_t xk = (uint32_t)((xi >> 20) & 0xFFFF);
                    const int st = gl == 0 ? 10 : gl == 1 ? 1 : gl == 2 ? 100 : 0;
                    const int d = (int)((xi >> 53) & 15);                // the position's own residue at the wildcard offset
                    uint32_t ok = 0;
                    if (COUNT) {
#pragma unroll
                        for (int j = 0; j < 10; j++) {
                            const int v = sd + (j - d) * st;             // st = 0 for the key group: the bucket stays
                            bool c = act && j != d;
                            if (c) { sc.lookups++; c = (bitmap[v >> 5] >> (v & 31)) & 1; }   // counting form: bucket occupancy decides, then the search
                            ok |= (uint32_t)c << j;
                        }
                    } else {   // pair filter: one 16-byte block answers for the ten residues (lanes without a pair read block 0)
                        const uint32_t hp = mc_pair_hash_d((uint32_t)sd, xk, gl, (uint32_t)d);
                        const uint4 blk = ((const uint4 *)X.pair)[act ? mc_pair_block(hp) : 0u];
                        ok = act ? (mc_pair_test4(blk.x, blk.y, blk.z, blk.w, hp) & ~(1u << d) & 0x3FFu) : 0u;
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
                    const uint32_t seed = (uint32_t)(pw & 0xFFFFF), qk = (uint32_t)(pw >> 20) & 0xFFFFu;
                    const uint32_t d3 = (uint32_t)(pw >> 50) & 15u, d4 = (uint32_t)(pw >> 54) & 15u, d5 = (uint32_t)(pw >> 58) & 15u;   // bucket digits at offsets 3, 4, 5
                    wdig = d4 | (d5 << 4) | (d3 << 8) | ((qk >> 12) << 12);  // the residue at the wildcard offset of groups 0..3
                    wbase = pw & 0x00007FFFFFFFFFFFull;                      // seed | key | position | frame: a queue item without its phase
                    if (pass == 0) {
                        const bool live0 = here && ((pw >> 47) & 1), live = here && ((pw >> 48) & 1);
                        bool defer = here && ((pw >> 49) & 1);
                        // both filters are asked before either answer is looked at: their reads are in flight together
                        const bool ask = live || defer;
                        const unsigned long long m9 = __ballot(live0), mw = __ballot(ask);
                        const bool any9 = !COUNT && m9, anyw = !COUNT && mw;
                        n_exact += (uint32_t)__popcll(m9); n_wild += (uint32_t)__popcll(mw);
                        const uint32_t qk0 = (qk & 0xFFF0u) | 0xFu;
                        uint32_t fw9 = 0, fb9 = 0, wsum = 0;
                        uint4 q0 = {0, 0, 0, 0}, q1 = {0, 0, 0, 0};
                        if (any9) {                                      // the exact 9-mer: its own Bloom filter, then straight into q
                            const uint32_t hh = mc_filter_hash(seed, qk0);
                            fb9 = mc_filter_bits(hh);
                            fw9 = X.filt[live0 ? mc_filter9_word(hh) : 0u];
                        }
                        if (anyw) {                                      // wildcard filter: one 32-byte line answers for the four groups
                            const uint32_t ctx = mc_wild_ctx(seed, qk);
                            const uint4 *ln = (const uint4 *)X.wild + (size_t)(ask ? mc_wild_line(ctx) : 0u) * 2;
                            q0 = ln[0]; q1 = ln[1];
                            wsum = mc_wild_sum(ctx, d3, d4, d5, qk >> 12);
                        }
                        const bool pr = live0 && (COUNT || (fw9 & fb9) == fb9);
                        const unsigned long long prm = __ballot(pr);
                        if (prm) {
                            if (pr) W->q[qn + __popcll(prm & lt)] = wbase | (0xFull << 20);   // phase 0; key g6 g7 g8 F
                            qn += __popcll(prm);
                        }
                        uint32_t wmt = 0xFu;                             // counting form: every probe is generated and searched
                        if (!COUNT) {
                            wmt = 0;
                            if (ask) wmt = (mc_wild_test2(q0.x, q0.y, mc_wild_bits_s(wsum, d4, 0)) ? 1u : 0u) | (mc_wild_test2(q0.z, q0.w, mc_wild_bits_s(wsum, d5, 1)) ? 2u : 0u) |
                                            (mc_wild_test2(q1.x, q1.y, mc_wild_bits_s(wsum, d3, 2)) ? 4u : 0u) | (mc_wild_test2(q1.z, q1.w, mc_wild_bits_s(wsum, qk >> 12, 3)) ? 8u : 0u);
                        }
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

// ---- best hits only (mc_set_best_hits_only) -------------------------------------------------------------------------------------
// classify_reads keeps, per read, the best-scoring m8 row that passes the family's three thresholds (microbe_census.py:432-460).
// A row is an HSP's own alignment (sum statistics only change its log E): a read none of whose HSPs would pass the thresholds as a
// row cannot be classified, whatever the ranking does - 99 % of the reads of a shotgun library.  The kernels that make HSPs mark
// the reads that have such an HSP (cand), and only THEIR HSPs - all of them: the others still decide the sums, the order and the
// 500-row cap - are sorted and finished.
#define MC_HSP_KEY(h) (((uint64_t)(h).read << 43) | ((uint64_t)(uint32_t)(h).sidx << 28) | (uint64_t)(h).chrono)   // (read, subject, hit order)
// frame and the four coordinates of an HSP in one word (3 + 8 + 8 + 11 + 11 bits: frames of up to 170 residues, markers of up to 1,192):
// two HSPs of a subject with the same word are one HSP found from several seeds (CalRes 0x4082b0-0x408446 keeps one of them)
// ... and above them the score (16 bits): of the HSPs of one place CalRes keeps the one with the smaller log E - the higher score, the
// first one found on a tie (every HSP's log E is still the table value of its score here: sum statistics come later)
#define MC_HSP_PLACE(h) (((uint64_t)(uint16_t)(h).score << 41) | ((uint64_t)(uint16_t)(h).frame << 38) | ((uint64_t)(uint16_t)(h).qaas << 30) | ((uint64_t)(uint16_t)(h).qaae << 22) | ((uint64_t)(uint16_t)(h).ds << 11) | (uint64_t)(uint16_t)(h).de)
#define MC_PLACE_OF(w) ((w) & ((1ull << 41) - 1))
#define MC_SCORE_OF(w) ((uint32_t)((w) >> 41))
__device__ __forceinline__ bool mc_hsp_can_classify(const McTables &T, const McClassPars &P, const McIndex &X, const int32_t *fam, const McHsp &h)
{
    const int f = fam[h.sidx];
    if (T.bits_r[h.score] < P.min_score[f]) return false;        // (most HSPs end here)
    McRow r;
    mc_fill_row(T, 0, h, r);
    return mc_row_passes(P, r, f, (int)(X.off[h.sidx + 1] - X.off[h.sidx]), r.frame);
}
// Seed hits -> HSPs / gap tasks.  Persistent workgroups walk the task pool 512 hits at a time; what survives is staged in
// LDS and flushed with ONE global atomic per ~400 HSPs / ~300 gap tasks: a device-scope atomic on a single counter executes
// at the memory side (the L2s of the XCDs are not coherent with each other) at ~125 M/s - one per HSP, or even one per wave,
// cost more than the whole evaluation (measured: 11.6 ms of which 7.9 ms atomics).
// The kernel waits on scattered byte reads of the residues (SQ_WAIT_ANY 77 % of the wave cycles), so it runs at the occupancy
// its registers allow, 24 waves per CU (80 VGPRs), as 3 workgroups of 8 waves whose staging pools just fit the LDS -
// measured per 1 M reads of 150 bp: 4 x 4 waves 5.3 ms, 4 x 5 waves 4.85, 3 x 8 waves 4.6, 2 x 12 waves 4.6; pools that flush
// more often (4 x 6 waves, 5 x 4 waves) 6.6 - 7.1.
// mc_eval_seed_tail (mc_core.h) for k_eval_seeds: the same growth, gate and ungapped X-drop extension, with the two extension
// loops reading EIGHT residues of both sequences per turn (one 8-byte load each, any alignment) and looking their eight scores
// up together - the plain loops make one trip to the L1 / L2 and one to LDS per residue, each waiting for the one before, and
// were half of the kernel's wave time (cycle counters).  The steps themselves are taken one residue at a time with the
// reference's exit tests, in the same order.  Rows and residue array have room on both sides (what a load reads past a
// sequence's end is never used: the step that would use it is behind an exit test).
__device__ __forceinline__ uint64_t mc_ld8(const uint8_t *p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
// growth and gate of a seed hit (mc_eval_seed_tail, mc_core.h): true if the hit goes on to the ungapped X-drop extension, with the
// grown seed (qp, dp, L), its score and identities
__device__ __forceinline__ bool mc_ev_gate(const McHot &T, const uint8_t *q, int qlen, int qpos, const uint8_t *d, int dlen, int dpos, int seedlen, int &score, int &ident,
                                           int &qp_o, int &dp_o, int &L_o, uint64_t q0, uint64_t q2, uint64_t d0, uint64_t d2)
{
    // growth: residues 9 .. 15 behind the seed's first one and the 8 in front of it are in registers (q2, d2 / q0, d0: the caller's
    // loads); most hits stop growing at once on both sides and reach the gate without another read
    int L = seedlen;
    int lim = dlen - dpos; if (lim > qlen - qpos) lim = qlen - qpos;
    // (seeds shorter than 9 residues - the generic seed kernel of a database whose .info threshold is above 0 emits 6 .. 9 - grow
    // residue by residue up to the ninth; the marker database's seeds are 9 or 10 long and never enter)
    while (L < 9 && lim > L && T.grp[q[qpos + L]] == T.grp[d[dpos + L]]) { int a = q[qpos + L], b = d[dpos + L]; score += MC_SUB(T, a, b); ident += (a == b); L++; }
#pragma unroll
    for (int j = 9; j < 16; j++) {
        const int a = (int)((q2 >> (8 * (j - 8))) & 0xFFu), b = (int)((d2 >> (8 * (j - 8))) & 0xFFu);
        if (L == j && lim > L && T.grp[a & 31] == T.grp[b & 31]) { score += MC_SUB(T, a, b); ident += (a == b); L++; }
    }
    if (L == 16) while (lim > L && T.grp[q[qpos + L]] == T.grp[d[dpos + L]]) { int a = q[qpos + L], b = d[dpos + L]; score += MC_SUB(T, a, b); ident += (a == b); L++; }
    int back = qpos < dpos ? qpos : dpos, qp = qpos, dp = dpos;
#pragma unroll
    for (int j = 1; j <= 8; j++) {
        const int a = (int)((q0 >> (8 * (8 - j))) & 0xFFu), b = (int)((d0 >> (8 * (8 - j))) & 0xFFu);
        if (qpos - qp == j - 1 && back > 0 && T.grp[a & 31] == T.grp[b & 31]) { qp--; dp--; back--; L++; score += MC_SUB(T, a, b); ident += (a == b); }
    }
    if (qpos - qp == 8) while (back > 0 && T.grp[q[qp - 1]] == T.grp[d[dp - 1]]) { qp--; dp--; back--; L++; int a = q[qp], b = d[dp]; score += MC_SUB(T, a, b); ident += (a == b); }
    qp_o = qp; dp_o = dp; L_o = L;
    return (double)score >= MC_SEED_SCORE && ident >= MC_SEED_IDENT;
}
// ... and the extension itself, from the grown seed: 1 = ungapped HSP complete, 2 = needs the gapped extension
__device__ __forceinline__ int mc_ev_xdrop(const McHot &T, const uint8_t *q, int qlen, const uint8_t *d, int dlen, int sidx, int qp, int dp, int L, int score, int ident, McGapTask *gt)
{
    const double xd = T.xdrop_ungapped;
    int s0 = score, qfwd = 0, qbwd = 0, fgain = 0, bgain = 0;
    { // forward
        const int n1 = qlen - qp - L, n2 = dlen - dp - L;
        int bl = 0, bi = 0;
        if (n1 != 0 && n2 != 0 && !(s0 < -20)) {
            const uint8_t *p1 = q + qp + L, *p2 = d + dp + L;
            int run = s0, best = s0, id = 0, i = 0;
            bool stop = false;
            do {
                const uint64_t wa = mc_ld8(p1 + i), wb = mc_ld8(p2 + i);
                int sc[8];
                uint32_t eq = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const uint32_t a = (uint32_t)(wa >> (8 * k)) & 0xFFu, b = (uint32_t)(wb >> (8 * k)) & 0xFFu;
                    sc[k] = (int)T.sub[((a << 5) | b) & 1023u]; eq |= (uint32_t)(a == b) << k;
                }
#pragma unroll
                for (int k = 0; k < 8; k++)
                    if (!stop) {
                        run += sc[k]; id += (int)((eq >> k) & 1u); i++;
                        if (run > best) { best = run; bl = i; bi = id; }
                        stop = !(n2 > i) || n1 <= i || run < -20 || (double)run < (double)best - xd;
                    }
            } while (!stop);
            fgain = best - s0;
        }
        ident += bi; qfwd = bl;
    }
    { // backward, restarting from the seed score
        int a = qp - 1, b = dp - 1, bl = 0, bi = 0;
        if (a >= 0 && b >= 0 && !(s0 < -20)) {
            int run = s0, best = s0, id = 0, cnt = 0;
            bool stop = false;
            do {
                const uint64_t wa = mc_ld8(q + a - 7), wb = mc_ld8(d + b - 7);       // residues a - 7 .. a: step k uses byte 7 - k
                int sc[8];
                uint32_t eq = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const uint32_t x = (uint32_t)(wa >> (8 * (7 - k))) & 0xFFu, y = (uint32_t)(wb >> (8 * (7 - k))) & 0xFFu;
                    sc[k] = (int)T.sub[((x << 5) | y) & 1023u]; eq |= (uint32_t)(x == y) << k;
                }
#pragma unroll
                for (int k = 0; k < 8; k++)
                    if (!stop) {
                        run += sc[k]; id += (int)((eq >> k) & 1u); cnt++;
                        if (best < run) { best = run; bl = cnt; bi = id; }
                        a--; b--;
                        stop = b < 0 || a < 0 || run < -20 || (double)run < (double)best - xd;
                    }
            } while (!stop);
            bgain = best - s0;
        }
        ident += bi; qbwd = bl;
    }
    score = s0 + bgain + fgain;
    gt->sidx = (uint32_t)sidx; gt->qp = (int16_t)qp; gt->dp = (int16_t)dp; gt->L = (int16_t)L;
    gt->qfwd = (int16_t)qfwd; gt->qbwd = (int16_t)qbwd; gt->score = (int16_t)score; gt->nmatch = (int16_t)ident;
    return (!(T.gap_trigger > (double)score)) ? 2 : 1;
}

#ifdef MC_EXP_TIMING
__device__ unsigned long long g_ev_acc[8];           // wave time per phase, summed over the waves: 0 barriers / flush 1 record, first reads, seed score 2 growth, gate, X-drop 3 HSP 4 staging
#define MC_EV_TICK(prev) do { const unsigned long long now_ = __builtin_readcyclecounter(); ev_acc_[prev] += now_ - ev_last_; ev_last_ = now_; } while (0)
#else
#define MC_EV_TICK(prev) do { } while (0)
#endif
#define MC_EV_BS 256         // threads per workgroup (the waves are on their own: the size only sets how the LDS is handed out)
#define MC_EV_BPC 5          // workgroups per CU: 20 waves, 5 per SIMD - 88 registers, nothing spilled (measured per 1 M reads of 150 / 300 bp:
                             // 7 waves per SIMD and 72 registers with 52 bytes of scratch 3.50 / 7.77 ms, 6 with 80 and 12 bytes 2.82 / 6.41, 5 with 88 2.60 / 6.01, 4: 2.79 / 6.53)
#define MC_EV_QCAP 128       // survivors of the gate a wave holds (32 bytes each: 4 KB of LDS per wave)
#define MC_EV_BLK 256u       // slots of the HSP / gap-task pools a wave reserves at a time (one global atomic per block)
// n consecutive slots for the wave's lanes (lane with rank r < n gets one; n is the same for every lane): from the wave's current
// block of the pool, continued in a new block when that one is full.  *ok = false after a pool overflow.
__device__ __forceinline__ uint32_t mc_ev_slots(uint32_t n, uint32_t r, uint32_t cap, uint32_t *counter, uint32_t &blk_base, uint32_t &blk_used, bool *ok, int lane)
{
    if (blk_used + n <= MC_EV_BLK) { const uint32_t s = blk_base + blk_used + r; blk_used += n; return s; }
    const uint32_t rem = MC_EV_BLK - blk_used, old = blk_base + blk_used;
    uint32_t nb = 0;
    if (lane == 0) nb = atomicAdd(counter, MC_EV_BLK);
    nb = (uint32_t)__builtin_amdgcn_readfirstlane((int)nb);
    if (nb + MC_EV_BLK > cap) { *ok = false; blk_used = MC_EV_BLK; return 0; }
    blk_base = nb; blk_used = n - rem;
    return r < rem ? old + r : nb + (r - rem);
}
// Seed hits -> HSPs / gap tasks, in two phases per wave.  Seven hits in ten end at the gate; the three that go on to the
// ungapped X-drop extension - long loops - used to do so in the lane that met them, 19 lanes of 64 on average.  Now a wave puts
// the survivors of the gate into a queue of its own in LDS (what the extension needs of them: 32 bytes) and runs the extension,
// the HSP and the classification mark on 64 survivors at a time - full waves.  No workgroup barrier in the loop and no staging
// pools: a record goes from its lane straight to the wave's current block of the global pool (blocks of 256 slots, one global
// atomic each; the records of a turn are consecutive, so the stores of the wave cover whole lines); what a wave does not use of
// its last block is padded with records the later stages skip (read = MC_TASK_NONE, sort key all ones; C_HPAD / C_GPAD count them).
__global__ void __attribute__((amdgpu_waves_per_eu(5, 5))) __launch_bounds__(MC_EV_BS) k_eval_seeds(const McTables *__restrict__ T, McIndex X, const uint8_t *__restrict__ frames, int FP, int L,
                                                    const McSeedTask *__restrict__ tasks, const uint32_t *__restrict__ ntasks_p, uint32_t cap_tasks, McHsp *hsps, uint32_t cap_hsps,
                                                    McGapTask *gaps, uint32_t cap_gaps, uint32_t *counters, const McClassPars *__restrict__ P, const int32_t *__restrict__ fam, uint8_t *cand, uint64_t *hkeys, uint8_t *low, uint64_t *hplace)
{
    const uint32_t ntasks = *ntasks_p <= cap_tasks ? *ntasks_p : 0u;   // (device-side count of the seed kernel; after an overflow the host discards the batch)
    __shared__ McHot hot;
    const int lane = mc_lane(), wv = threadIdx.x >> 6;
    uint4 *Q = (uint4 *)(mc_smem + (size_t)wv * MC_EV_QCAP * 32);    // entry e: words 2 e, 2 e + 1
    mc_load_hot(&hot, T);
    __syncthreads();
    const double hot_loge_thr = T->loge_thr;
    uint32_t qn = 0, hb_base = 0, hb_used = MC_EV_BLK, gb_base = 0, gb_used = MC_EV_BLK;
    bool ok = true;
    const unsigned long long lt = (1ull << lane) - 1;
    const uint32_t nchunks = (ntasks + MC_EV_BS - 1) / MC_EV_BS;
    // The chain of dependent reads of a hit was: its record -> the subject's offsets -> the residue in front of the seed -> the
    // seed's residues, four trips to the L2 before the gate.  Now: the record of the NEXT chunk is fetched while this one is
    // evaluated, the record carries the hit's position in the residue array (MC_TASK_W3), and the subject's end, the residues in
    // front of the seed and the seed's own ten are read together: one trip.
#ifdef MC_EXP_TIMING
    unsigned long long ev_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ev_last_ = __builtin_readcyclecounter();
#endif
    McSeedTask tn;
    tn.read = MC_TASK_NONE; tn.chrono = 0; tn.posting = 0; tn.seedlen_nkey = 0;
    if (blockIdx.x * MC_EV_BS + threadIdx.x < ntasks) tn = tasks[blockIdx.x * MC_EV_BS + threadIdx.x];
    for (uint32_t chunk = blockIdx.x;; chunk += gridDim.x) {
        const bool last = chunk >= nchunks;
        if (!last) {   // ---- phase 1: the gate, one hit per lane
            MC_EV_TICK(0);
            const uint32_t tid = chunk * MC_EV_BS + threadIdx.x;
            const McSeedTask t = tn;
            {
                const uint64_t nx = (uint64_t)(chunk + gridDim.x) * MC_EV_BS + threadIdx.x;
                tn.read = MC_TASK_NONE;
                if (nx < ntasks) tn = tasks[nx];
            }
            bool surv = false;
            uint4 e0 = {0, 0, 0, 0}, e1 = {0, 0, 0, 0};
            if (tid < ntasks && t.read != MC_TASK_NONE) {            // (MC_TASK_NONE: padding of a partly used block of the task pool)
                const int frame = (int)(t.chrono >> 25), pos = (int)((t.chrono >> 17) & 0xff);
                const int qlen = (L - frame % 3) / 3;
                const uint32_t w3 = t.seedlen_nkey;
                const int seedlen = (int)((w3 >> 24) & 15u), nkey = (int)(w3 >> 28), dpos = (int)(t.posting & 0x7ffu), sidx = (int)(t.posting >> 11);
                const uint32_t o0 = (w3 & 0xFFFFFFu) - (uint32_t)dpos;
                const uint8_t *q = frames + ((int64_t)t.read * 6 + frame) * FP, *d = X.res + o0;
                const uint32_t o1 = X.off[sidx + 1];
                // residues pos - 8 .. pos + 15 of the frame and dpos - 8 .. dpos + 15 of the subject: six loads, one trip (rows and residue array have room on both sides)
                const uint64_t q0 = mc_ld8(q + pos - 8), q1 = mc_ld8(q + pos), q2 = mc_ld8(q + pos + 8), d0 = mc_ld8(d + dpos - 8), d1 = mc_ld8(d + dpos), d2 = mc_ld8(d + dpos + 8);
                const int qm1 = (int)(q0 >> 56), dm1 = (int)(d0 >> 56);
                const int dlen = (int)(o1 - o0);
                int score = 0, ident = 0;
#pragma unroll
                for (int k = 0; k < 10; k++)
                    if (k < seedlen) {
                        const int a = (int)((k < 8 ? q1 >> (8 * k) : q2 >> (8 * (k - 8))) & 0xFFu), b = (int)((k < 8 ? d1 >> (8 * k) : d2 >> (8 * (k - 8))) & 0xFFu);
                        score += MC_SUB(hot, a, b); ident += (a == b);
                    }
                const bool go = !(dpos + seedlen > dlen) && !(pos != 0 && dpos != 0 && hot.grp[qm1] == hot.grp[dm1] && nkey != 4);
                int qp = 0, dp = 0, Lg = 0;
                if (go) surv = mc_ev_gate(hot, q, qlen, pos, d, dlen, dpos, seedlen, score, ident, qp, dp, Lg, q0, q2, d0, d2);
                e0.x = t.read; e0.y = t.chrono; e0.z = o0; e0.w = (uint32_t)sidx;
                e1.x = (uint32_t)qp | ((uint32_t)dp << 16); e1.y = (uint32_t)Lg | ((uint32_t)(uint16_t)(int16_t)score << 16); e1.z = (uint32_t)ident | ((uint32_t)dlen << 16);
            }
            const unsigned long long ms = __ballot(surv);
            if (surv) { const uint32_t at = qn + (uint32_t)__popcll(ms & lt); Q[2 * at] = e0; Q[2 * at + 1] = e1; }
            qn += (uint32_t)__popcll(ms);
            mc_wave_sync();
            MC_EV_TICK(1);
        }
        while (qn >= 64 || (last && qn > 0)) {   // ---- phase 2: the extension, 64 survivors at a time
            const uint32_t take = qn < 64 ? qn : 64;
            qn -= take;
            const bool act = (uint32_t)lane < take;
            const uint4 e0 = Q[2 * (qn + (act ? (uint32_t)lane : 0u))], e1 = Q[2 * (qn + (act ? (uint32_t)lane : 0u)) + 1];
            mc_wave_sync();                                          // (read before the next survivors are written over them)
            int rc = 0;
            bool keep = false;
            McGapTask g;
            McHsp h;
            if (act) {
                const uint32_t read = e0.x, chrono = e0.y;
                const int frame = (int)(chrono >> 25), qlen = (L - frame % 3) / 3, sidx = (int)e0.w;
                const uint8_t *q = frames + ((int64_t)read * 6 + frame) * FP, *d = X.res + e0.z;
                g.read = read; g.chrono = chrono;
                rc = mc_ev_xdrop(hot, q, qlen, d, (int)(e1.z >> 16), sidx, (int)(e1.x & 0xFFFFu), (int)(e1.x >> 16), (int)(e1.y & 0xFFFFu), (int)(int16_t)(e1.y >> 16), (int)(e1.z & 0xFFFFu), &g);
                MC_EV_TICK(2);
                if (rc == 1) {
                    h.read = read; h.chrono = chrono;
                    keep = mc_make_hsp(*T, L, frame, g, g.qfwd, g.qfwd, g.qbwd, g.qbwd, g.score, g.nmatch, g.qfwd + g.L + g.qbwd, 0, 0, &h);
                    if (keep && cand && mc_hsp_can_classify(*T, *P, X, fam, h)) cand[h.read] = 1;
                    if (keep && h.loge < hot_loge_thr) low[h.read] = 1;      // (the read can print a row: k_order_light)
                }
            }
            MC_EV_TICK(3);
            const unsigned long long mh = __ballot(keep), mg = __ballot(rc == 2);
            if (mh && ok) {
                const uint32_t slot = mc_ev_slots((uint32_t)__popcll(mh), (uint32_t)__popcll(mh & lt), cap_hsps, &counters[C_HSPS], hb_base, hb_used, &ok, lane);
                if (!ok) { if (lane == 0) counters[C_OVERFLOW] = 2; }
                else if (keep) { hsps[slot] = h; hkeys[slot] = MC_HSP_KEY(h); hplace[slot] = MC_HSP_PLACE(h); }
            }
            if (mg && ok) {
                const uint32_t slot = mc_ev_slots((uint32_t)__popcll(mg), (uint32_t)__popcll(mg & lt), cap_gaps, &counters[C_GAPS], gb_base, gb_used, &ok, lane);
                if (!ok) { if (lane == 0) counters[C_OVERFLOW] = 3; }
                else if (rc == 2) gaps[slot] = g;
            }
            MC_EV_TICK(4);
        }
        if (last) break;
    }
    {   // what the wave did not use of its last blocks: records the later stages skip
        const uint32_t ph = hb_used < MC_EV_BLK ? MC_EV_BLK - hb_used : 0u, pg = gb_used < MC_EV_BLK ? MC_EV_BLK - gb_used : 0u;
        if (ok) {
            for (uint32_t i = (uint32_t)lane; i < ph; i += 64) { hsps[hb_base + hb_used + i].read = MC_TASK_NONE; hkeys[hb_base + hb_used + i] = ~0ull; }
            for (uint32_t i = (uint32_t)lane; i < pg; i += 64) gaps[gb_base + gb_used + i].read = MC_TASK_NONE;
            if (lane == 0) { if (ph) atomicAdd(&counters[C_HPAD], ph); if (pg) atomicAdd(&counters[C_GPAD], pg); }
        }
    }
#ifdef MC_EXP_TIMING
    if (lane == 0) for (int k = 0; k < 5; k++) atomicAdd(&g_ev_acc[k], ev_acc_[k]);
#endif
}

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
// their sort keys (1 + DP rows; the slots behind the list keep key 0 from the memset and sort to the end)
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
                                                   uint32_t *retry_count, uint32_t *retry, int refill)
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
    const uint32_t share = nitems > w0 ? (nitems - w0 + G - 1) / G : 0u;   // items of this wave: list[w0 + G k], k < share
    const unsigned long long lt = (1ull << lane) - 1;
    const int REFILL = LANES < refill ? 1 : refill;
    uint32_t taken = 0;
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
            if (rm && (__popcll(rm) >= REFILL || taken >= share || __ballot(active) == 0)) {
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
            if (cm && taken < share) {
                const uint32_t k = taken + (uint32_t)__popcll(cm & lt);
                if (want && k < share) { nit = list[w0 + G * k]; nstage = 1; }
                taken += (uint32_t)__popcll(cm);
            }
        }
        // ---- one DP row of every flank in progress
        if (active && !fin) fin = mc_gap_row(hot, S, ws, W);
        if (fin && ws.ovf) S.over = 1;                              // (more than 31 gap runs on a live path: redone with the wider launch, in the end with full-size cells)
        if (fin) { fout[it] = mc_flank_out(mc_gap_result(S)); active = false; }
        const bool over = fin && S.over != 0;
        const uint32_t ro = mc_wave_alloc(retry_count, over);      // band left the window: the flank is redone with a wider one
        if (over) retry[ro] = it;
        if (taken >= share && __ballot(active || nstage != 0) == 0) break;
    }
}

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
#define MC_ORDER_SMALL 512                 // segments up to this long: a wave per read (two buffers of 4 KB in LDS) ...
#define MC_ORDER_MID 2048                  // ... up to this long (3 reads in 1,000): a workgroup of four waves (two buffers of 16 KB) ...
#define MC_ORDER_LDS 8192                  // ... the few longer ones (0.6 in 1,000): a workgroup of sixteen waves (two buffers of 64 KB; beyond that: blocks of 8192, merged in global memory)
// the reads whose segments are longer than MC_BIN_LIGHT, listed for k_order_heavy (a thread per read)
__global__ void __launch_bounds__(256) k_order_lists(const uint32_t *__restrict__ heads, uint32_t nreads, uint32_t *counters, uint32_t *heavy, uint32_t *heavy2, uint32_t *heavy3)
{
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    const uint32_t n = r < nreads ? heads[r + 1] - heads[r] : 0u;
    const bool c1 = n > MC_BIN_LIGHT && n <= MC_ORDER_SMALL, c2 = n > MC_ORDER_SMALL && n <= MC_ORDER_MID, c3 = n > MC_ORDER_MID;
    const uint32_t o = mc_block_alloc(&counters[C_ORDER], c1);
    if (c1) heavy[o] = r;
    const uint32_t o2 = mc_block_alloc(&counters[C_ORDER2], c2);
    if (c2) heavy2[o2] = r;
    const uint32_t o3 = mc_block_alloc(&counters[C_ORDER3], c3);
    if (c3) heavy3[o3] = r;
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
    // ... and a thread per marked read walks its HSPs in order: CalRes' stacks (mc_build_stacks, mc_finish.h) - of the consecutive
    // HSPs of one place the best one, the subject's stack newest first, its size with the first record
    if (threadIdx.x < MC_OL_READS && lmark[threadIdx.x]) {
        const uint32_t q = threadIdx.x, n = lcnt[q], base = lpos[q], a = lhead[q];
        uint32_t out = 0;
        for (uint32_t gs = 0; gs < n;) {
            const uint64_t sx = key[base + gs] >> 28;
            uint32_t ge = gs + 1, kg = 1;
            while (ge < n && (key[base + ge] >> 28) == sx) { kg += MC_PLACE_OF(plc[base + ge]) != MC_PLACE_OF(plc[base + ge - 1]) ? 1u : 0u; ge++; }
            uint32_t run = 0;
            for (uint32_t j = gs; j < ge; run++) {
                uint32_t bestj = j, j2 = j + 1;
                while (j2 < ge && MC_PLACE_OF(plc[base + j2]) == MC_PLACE_OF(plc[base + j])) { if (MC_SCORE_OF(plc[base + j2]) > MC_SCORE_OF(plc[base + bestj])) bestj = j2; j2++; }
                const uint32_t o = a + out + kg - 1 - run;
                order[o] = slots[a + sid[base + bestj]];
                gsz[o] = run == kg - 1 ? kg : 0u;
                j = j2;
            }
            out += kg; gs = ge;
        }
        nv[r0 + q] = out;
        nrow_of[r0 + q] = 1u;
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

#ifndef MC_FH_MIN
#define MC_FH_MIN 96
#endif
#ifndef MC_FH_MIN_BEST
#define MC_FH_MIN_BEST 32
#endif
// MC_FH_MIN: reads with more stacked HSPs than this are finished by a whole wave (k_finish_heavy); MC_FH_MIN_BEST: the same with
// best hits only, where few reads are finished and the longest thread of k_finish decides (per 1 M reads of 150 bp: 96 / 48 / 32 / 16
// -> finishing 2.36 / 2.64 / 2.65 / 3.80 ms with rows, 2.38 / 1.43 / 1.41 / 1.42 ms with best hits only)
#define MC_FH_N1 512      // subjects / ranked HSPs a read may have in the first wave kernel (11 KB of LDS per wave) ...
#define MC_FH_N2 2048     // ... in the second (45 KB) ...
#define MC_FH_N3 6144     // ... and in the third (135 KB, one wave per CU), where anything larger is finished by lane 0 alone

// The reads that get a wave of their own (k_finish_heavy), collected before the finishing kernels start so that they can run
// beside the thread-per-read kernel on a second stream.  A read without a marked HSP prints nothing whatever its size.
// The reads with a marked HSP that a single thread finishes (k_finish) are listed by size class as well: a wave of k_finish
// then holds reads of similar size instead of one read of 90 HSPs among 63 idle lanes.
#define MC_LIGHT_CLASS(n) ((n) <= 4 ? 0 : (n) <= 16 ? 1 : (n) <= 48 ? 2 : 3)
__global__ void __launch_bounds__(256) k_heavy_lists(const uint32_t *__restrict__ nv, uint32_t nheads, uint32_t *nrow_of,
                                                     McBestHit *best_of, uint32_t *counters, uint32_t *heavy, uint32_t *light, uint32_t light_pitch, uint32_t fh_min)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    int cls = -1;                                                // -1 nothing to do, 0..3 light class, 4 heavy
    if (s < nheads) {
        const uint32_t any = nrow_of[s];                           // (k_order_*: the read is marked, nv[s] = the size of its stacks)
        if (!any) best_of[s].family = -1;
        else { const uint32_t n = nv[s]; cls = n > fh_min ? 4 : MC_LIGHT_CLASS(n); }
    }
    const uint32_t o = mc_block_alloc(&counters[C_HEAVY], cls == 4);
    if (cls == 4) heavy[o] = s;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const uint32_t oc = mc_block_alloc(&counters[C_LIGHT0 + c], cls == c);
        if (cls == c) light[(size_t)c * light_pitch + oc] = s;
    }
}

// One thread per marked read.  All scratch is addressed by the read's offset into the binned HSPs (heads; a read never produces
// more rows than it has HSPs): v = the stacks (built by the ordering kernels), tmp = 2 HSP slots per HSP for the
// sum statistics, reused afterwards for the read's rows and their merge keys (64 + 8 bytes per row <= 96).  The rows
// stay in that scratch; k_emit_rows moves them to their final place once the row counts have been scanned.
__global__ void __launch_bounds__(256) k_finish(const McTables *__restrict__ T, McIndex X, const McClassPars *__restrict__ P, const int32_t *__restrict__ fam,
                                                const uint32_t *__restrict__ nv, const uint32_t *__restrict__ heads, uint32_t nheads,
                                                McHsp *v, McHsp *tmp, int64_t first_read_id, uint32_t *nrow_of, McBestHit *best,
                                                const uint32_t *__restrict__ light, uint32_t light_pitch, const uint32_t *__restrict__ nlight)
{
    // blockIdx.y = size class, the largest first: the four classes in ONE launch - a thread walks its read alone at the latency of
    // global memory and the reads that print anything fill a fraction of the GPU, so four launches one after the other took four
    // times the slowest thread of a class
    const int cl = 3 - (int)blockIdx.y;
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nlight[cl]) return;                                // (the reads of one size class that have something to print: k_heavy_lists)
    const uint32_t s = light[(size_t)cl * light_pitch + idx];     // the read; its stacks: v[heads[s] ...], nv[s] records (k_order_*)
    const uint32_t a = heads[s];
    const int n = (int)(heads[s + 1] - a);                        // (the scratch of a read is laid out by the size of its segment)
    McRow *myrows = (McRow *)(tmp + 2 * (size_t)a);
    double *myk = (double *)(myrows + n);
    McBestHit bh;
    McSortItem *myitems = (McSortItem *)(myk + n);               // 64 n + 8 n + 16 n = 88 n <= 96 n bytes of the read's tmp area
    const int nr = mc_finish_stacked(*T, X, *P, fam, (int)((int64_t)s + first_read_id), v + a, (int)nv[s], tmp + 2 * (size_t)a, myrows, myk, myitems, &bh);
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
#define MC_HL_H(e) lds[((e) << 6) + lane]
__global__ void __launch_bounds__(64) k_heap_lanes(const uint32_t *__restrict__ heads, uint32_t nheads, uint32_t nhsps, McHsp *tmp, const uint32_t *__restrict__ nrow_of,
                                                   const uint32_t *__restrict__ counters, const uint32_t *__restrict__ heavy_first)
{
    uint32_t *lds = (uint32_t *)mc_smem;
    __shared__ uint32_t s_nr[64];
    __shared__ uint32_t *s_hw[64];
    const int lane = mc_lane();
    const uint32_t nheavy = counters[C_HEAVY];
    for (uint32_t slot0 = blockIdx.x * 64u; slot0 < nheavy; slot0 += gridDim.x * 64u) {
        int n = 0;
        {
            const uint32_t slot = slot0 + (uint32_t)lane;
            uint32_t *hw = nullptr;
            if (slot < nheavy) {
                const uint32_t e = heavy_first[slot];
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
        if (n >= 2) {   // mc_heapsort (mc_sort_impl.h) move for move, element e in word e + 1; the keys are the upper halves
#define MC_HL_ADJUST(HOLE, LEN, VALUE)                                                                                             \
    do {                                                                                                                           \
        int hole_ = (HOLE), sc_ = hole_;                                                                                           \
        const int top_ = hole_, len_ = (LEN);                                                                                      \
        const uint32_t value_ = (VALUE);                                                                                           \
        while (sc_ < (len_ - 1) / 2) {                                                                                             \
            sc_ = 2 * (sc_ + 1);                                                                                                   \
            const uint32_t cx_ = MC_HL_H(sc_), cy_ = MC_HL_H(sc_ + 1);            /* elements sc - 1 and sc */                      \
            uint32_t pick_ = cy_;                                                                                                  \
            if ((cy_ >> 16) < (cx_ >> 16)) { sc_--; pick_ = cx_; }                                                                 \
            MC_HL_H(hole_ + 1) = pick_; hole_ = sc_;                                                                               \
        }                                                                                                                          \
        if ((len_ & 1) == 0 && sc_ == (len_ - 2) / 2) { sc_ = 2 * (sc_ + 1); MC_HL_H(hole_ + 1) = MC_HL_H(sc_); hole_ = sc_ - 1; } \
        int parent_ = (hole_ - 1) / 2;                                                                                             \
        while (hole_ > top_ && (MC_HL_H(parent_ + 1) >> 16) < (value_ >> 16)) { MC_HL_H(hole_ + 1) = MC_HL_H(parent_ + 1); hole_ = parent_; parent_ = (hole_ - 1) / 2; } \
        MC_HL_H(hole_ + 1) = value_;                                                                                               \
    } while (0)
            for (int parent = (n - 2) / 2;; parent--) { MC_HL_ADJUST(parent, n, MC_HL_H(parent + 1)); if (parent == 0) break; }
            for (int m = n; m > 1;) { m--; const uint32_t vv = MC_HL_H(m + 1); MC_HL_H(m + 1) = MC_HL_H(1); MC_HL_ADJUST(0, m, vv); }
#undef MC_HL_ADJUST
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
#undef MC_HL_H

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
        MC_FH_TICK(0);
        const uint32_t slot = CTR == C_HEAVY ? bi : list[bi];        // position in the first list (heavy_first): the later lists hold slots
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
    (void)mc_block_alloc(&counters[C_SEGS], nr > 0);
    const uint32_t o = mc_block_alloc(&counters[C_BEST], bh.family >= 0);
    if (bh.family >= 0) best[o] = bh;
}

// ---- the training workflow's grid search over the classification parameters (training/training.py:311-334) -------------------
// classify_reads there filters the m8 rows by (aln_cov, max_pid, min_score), keeps the best-scoring row per read (the first on
// a tie) and counts hits / aligned residues / coverage per family - for every combination of 4 x 6 x 27 parameter values, one
// pass over the file each.  Here one thread per read does all of it in one pass over the read's rows: for a given (aln_cov,
// max_pid) the best row does not depend on min_score (a higher cut-off only removes lower rows), so the read contributes its
// best row to every cut-off <= that row's bit score - one atomic into bin k = number of (ascending) cut-offs it reaches; the
// host turns the bins into the per-cut-off counts with a suffix sum.
#define MC_GRID_MAXC 8
#define MC_GRID_MAXP 8
#define MC_GRID_MAXS 64
struct McGridPars { int read_len, n_cov, n_pid, n_score, nfam; double cov[MC_GRID_MAXC]; int pid[MC_GRID_MAXP]; double score[MC_GRID_MAXS]; };

__global__ void __launch_bounds__(128) k_grid_classify(McGridPars G, McIndex X, const int32_t *__restrict__ fam, const McRow *__restrict__ rows, int64_t nrows,
                                                       unsigned long long *bin_hits, unsigned long long *bin_aln, double *bin_cov)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows) return;
    const int q = rows[i].query;
    if (i > 0 && rows[i - 1].query == q) return;                 // one thread per read: the one at its first row
    double bbits[MC_GRID_MAXC * MC_GRID_MAXP];
    int bidx[MC_GRID_MAXC * MC_GRID_MAXP];
    for (int c = 0; c < G.n_cov * G.n_pid; c++) { bbits[c] = 0.0; bidx[c] = -1; }
    for (int64_t k = i; k < nrows && rows[k].query == q; k++) {
        const McRow r = rows[k];
        const int tl = (int)(X.off[r.subject + 1] - X.off[r.subject]);
        const double cov = mc_row_coverage(G.read_len, r, tl);
        for (int ic = 0; ic < G.n_cov; ic++) {
            if (cov < G.cov[ic]) continue;
            for (int ip = 0; ip < G.n_pid; ip++) {
                if (100 * r.frame > G.pid[ip] * r.alnlen) continue;          // pid > max_pid (McRow::frame carries the identities)
                const int c = ic * G.n_pid + ip;
                if (bidx[c] < 0 || bbits[c] < r.bits) { bbits[c] = r.bits; bidx[c] = (int)(k - i); }
            }
        }
    }
    for (int c = 0; c < G.n_cov * G.n_pid; c++) {
        if (bidx[c] < 0) continue;
        int nk = 0;
        for (int j = 0; j < G.n_score; j++) nk += !(bbits[c] < G.score[j]) ? 1 : 0;   // cut-offs ascending: the row passes the first nk of them
        if (nk == 0) continue;
        const McRow r = rows[i + bidx[c]];
        const int f = fam[r.subject], tl = (int)(X.off[r.subject + 1] - X.off[r.subject]);
        const size_t o = ((size_t)c * (MC_GRID_MAXS + 1) + (size_t)nk) * (size_t)G.nfam + (size_t)f;
        atomicAdd(&bin_hits[o], 1ull);
        atomicAdd(&bin_aln[o], (unsigned long long)r.alnlen);
        atomicAdd(&bin_cov[o], (double)r.alnlen / (double)tl);
    }
}

// ------------------------------------------------------------------------------------------------
// handle
// ------------------------------------------------------------------------------------------------
// Everything one batch in flight needs.  A handle owns MC_NCTX of them: mc_run_range() cuts its range into as many parts and
// issues their stages alternately, so that while the host waits for the counters of one part the GPU works on the other
// (and the tails of the latency-bound kernels of one part overlap the kernels of the other).
#define MC_NCTX 2
struct McCtx {
    hipStream_t stream = nullptr, side = nullptr;
    hipEvent_t ev[8] = {}, ev_fork = nullptr, ev_join = nullptr;
    int64_t cap_reads = 0;
    uint32_t cap_tasks = 0, cap_gaps = 0, cap_hsps = 0, cap_rows = 0;
    uint8_t *d_frames = nullptr, *d_frames_base = nullptr;   // (64 bytes of room in front: k_eval_seeds reads 8 bytes at a time backwards from a seed)
    unsigned long long *d_stats = nullptr;
    McSeedTask *d_tasks = nullptr; McGapTask *d_gaps = nullptr; McHsp *d_hsps = nullptr, *d_v = nullptr, *d_tmp = nullptr;
    uint64_t *d_k64 = nullptr, *d_hkeys = nullptr, *d_hplace = nullptr, *d_places = nullptr; uint32_t *d_idx = nullptr, *d_idxo = nullptr, *d_heads = nullptr, *d_scan = nullptr, *d_gsz = nullptr, *d_nv = nullptr; void *d_sorttmp = nullptr; size_t sorttmp_bytes = 0;
    uint32_t *d_counters = nullptr;
    McRow *d_rows = nullptr; uint32_t *d_nrow = nullptr, *d_rowoff = nullptr; McBestHit *d_best = nullptr, *d_bestof = nullptr; uint8_t *d_low = nullptr, *d_cand = nullptr;
    McGapCell *d_gws_full = nullptr; uint32_t *d_retry = nullptr, *d_retry2 = nullptr; int gap_threads_full = 0;
    unsigned long long *d_gtab = nullptr; uint32_t gtab_slots = 0; uint32_t *d_gleader = nullptr; McFlankOut *d_fout = nullptr;
    // pinned host mirrors
    uint32_t *h_c = nullptr; unsigned long long *h_stats = nullptr; McBestHit *h_best = nullptr; size_t h_best_cap = 0;
    // the part being processed
    const uint8_t *reads = nullptr; int64_t n = 0, first_read_id = 0;
    uint32_t ntasks = 0, ngaps = 0, gpad = 0, nh = 0, nh_all = 0, nheads = 0, nrows = 0, nbest = 0, nsegs = 0;
};

struct mc_handle {
    McHostIndex H;
    std::vector<int32_t> fam;
    int nfam = 0, device = 0;
    // device index + tables
    uint8_t *d_res = nullptr, *d_res_base = nullptr; uint32_t *d_off = nullptr, *d_bstart = nullptr, *d_post = nullptr; uint16_t *d_keys = nullptr; int32_t *d_fam = nullptr;
    McTables *d_T = nullptr; McClassPars *d_P = nullptr;
    McTables hT; McClassPars hP;
    int read_len = 0, FP = 0; bool run_set = false;
    uint32_t *d_bitmap = nullptr;
    McBucketRec *d_rec = nullptr;
    uint32_t *d_filt = nullptr, *d_wild = nullptr, *d_pair = nullptr; uint64_t *d_segtab = nullptr; unsigned long long *d_rt = nullptr;
    bool fast_enum = false;
    bool count_traffic = false;
    int parts = 1;                        // parts a range is cut into (mc_set_parts): 1 = one kernel at a time; 2 = two halves whose stages alternate
    bool keep_rows = true;                // mc_search / mc_search_files hand out the m8 rows (mc_set_keep_rows)
    bool best_only = false;               // only the reads that can be classified are ranked; no rows (mc_set_best_hits_only)
    uint8_t *stage_pin[2] = {nullptr, nullptr}, *stage_dev[2] = {nullptr, nullptr}; size_t stage_bytes = 0; hipStream_t copy_stream = nullptr;   // run_stream
    // resident reads
    int64_t nreads = 0, cap_own = 0;
    uint8_t *d_reads = nullptr;
    const uint8_t *reads_dev = nullptr;   // resident read set (own buffer or attached caller memory)
    McCtx ctx[MC_NCTX];
    // host results: rows of the last run land in pinned memory; mc_search() accumulates its batches in all_rows
    // The rows travel to the host while the caller goes on (two pinned buffers in turn, a stream and an event of their own):
    // mc_run_range() returns when the best hits are there; whoever looks at the rows waits for their copy (rows_wait).
    mc_row *pin_slot[2] = {nullptr, nullptr}; size_t pin_slot_cap[2] = {0, 0}; int pin_cur = 0;
    mc_row *pin_rows = nullptr; size_t pin_cap = 0;                // the slot of the current run
    hipStream_t rows_stream = nullptr; hipEvent_t ev_rows = nullptr; bool rows_pending = false, rows_ever = false;
    std::vector<mc_row> all_rows, split_rows;                       // accumulated over the batches of a stream / over the halves of a range that overflowed
    const mc_row *res_rows = nullptr; int64_t n_res_rows = 0;
    std::vector<mc_best_hit> best; mc_stats stats;
};

static McIndex dev_index(const mc_handle *h)
{
    McIndex X; X.res = h->d_res; X.off = h->d_off; X.bstart = h->d_bstart; X.post = h->d_post; X.keys = h->d_keys; X.rec = h->d_rec; X.filt = h->d_filt; X.wild = h->d_wild; X.pair = h->d_pair; X.rt = h->d_rt; X.rt_mask = h->H.rt_mask; X.nseq = h->H.nseq;
    return X;
}

template <class Tp> static int dalloc(Tp **p, size_t n)
{
    if (*p) { (void)hipFree(*p); *p = nullptr; }
    HIPCK(hipMalloc((void **)p, n * sizeof(Tp)));
    return 0;
}

extern "C" int mc_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static void ctx_free(McCtx &c)
{
    void *ptrs[] = {c.d_frames_base, c.d_tasks, c.d_gaps, c.d_hsps, c.d_v, c.d_tmp, c.d_k64, c.d_hkeys, c.d_hplace, c.d_places, c.d_idx, c.d_idxo, c.d_heads, c.d_scan, c.d_gsz, c.d_nv, c.d_sorttmp, c.d_counters, c.d_rows,
                    c.d_nrow, c.d_rowoff, c.d_best, c.d_bestof, c.d_low, c.d_cand, c.d_gws_full, c.d_retry, c.d_retry2, c.d_gtab, c.d_gleader, c.d_fout, c.d_stats};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    for (void *p : {(void *)c.h_c, (void *)c.h_stats, (void *)c.h_best}) if (p) (void)hipHostFree(p);
    for (auto &e : c.ev) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : {c.ev_fork, c.ev_join}) if (e) (void)hipEventDestroy(e);
    for (hipStream_t q : {c.stream, c.side}) if (q) (void)hipStreamDestroy(q);
    c = McCtx();
}

extern "C" void mc_close(mc_handle *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    void *ptrs[] = {h->d_res_base, h->d_off, h->d_bstart, h->d_post, h->d_keys, h->d_fam, h->d_T, h->d_P, h->d_reads, h->d_bitmap, h->d_rec, h->d_filt, h->d_wild, h->d_pair, h->d_rt, h->d_segtab};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    for (McCtx &c : h->ctx) ctx_free(c);
    for (int k = 0; k < 2; k++) { if (h->stage_pin[k]) (void)hipHostFree(h->stage_pin[k]); if (h->stage_dev[k]) (void)hipFree(h->stage_dev[k]); }
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    if (h->rows_stream) { (void)hipStreamSynchronize(h->rows_stream); (void)hipStreamDestroy(h->rows_stream); }
    if (h->ev_rows) (void)hipEventDestroy(h->ev_rows);
    for (mc_row *p : h->pin_slot) if (p) (void)hipHostFree(p);
    delete h;
}

// h->H holds the host index (built from FASTA or loaded from a rapdb): everything device side
static int open_impl(mc_handle *h, const int32_t *marker_family, int32_t nfam, int32_t device)
{
    int ndev = 0;
    double t0 = mc_now();
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_err = "no HIP device available: libmcensus_hip has no CPU fallback"; return -1; }
    if (device < 0 || device >= ndev) { g_err = "device index out of range"; return -1; }
    if (nfam > 32) { g_err = "at most 32 gene families are supported"; return -1; }
    const int nseq = h->H.nseq;
    if (nseq > 32767) { g_err = "more than 32767 markers: the HSP sort key (read<<43 | subject<<28 | hit order) holds 15 bits of subject index"; return -1; }
    for (int s = 0; s < nseq; s++) if ((int)(h->H.off[s + 1] - h->H.off[s]) > MC_GAP_W - 8) { g_err = "marker longer than the gapped-extension workspace"; return -1; }
    if (marker_family) h->fam.assign(marker_family, marker_family + nseq); else h->fam.assign((size_t)nseq, 0);
    h->nfam = nfam; h->device = device;
    HIPCK(hipSetDevice(device));
    HIPCK(hipFree(nullptr));
    MC_OT("  HIP runtime, device", t0);
    for (McCtx &c : h->ctx) {
        HIPCK(hipStreamCreate(&c.stream)); HIPCK(hipStreamCreate(&c.side));
        for (auto &e : c.ev) HIPCK(hipEventCreate(&e));
        HIPCK(hipEventCreateWithFlags(&c.ev_fork, hipEventDisableTiming)); HIPCK(hipEventCreateWithFlags(&c.ev_join, hipEventDisableTiming));
        if (dalloc(&c.d_counters, C_N) || dalloc(&c.d_stats, S_N)) return -1;
        HIPCK(hipHostMalloc((void **)&c.h_c, sizeof(uint32_t) * C_N, hipHostMallocDefault));
        HIPCK(hipHostMalloc((void **)&c.h_stats, sizeof(unsigned long long) * S_N, hipHostMallocDefault));
    }
    HIPCK(hipStreamCreate(&h->rows_stream)); HIPCK(hipEventCreateWithFlags(&h->ev_rows, hipEventDisableTiming));
    const McHostIndex &H = h->H;
    if (H.res.size() >= MC_TASK_ABS_LIMIT) { g_err = "marker database too large: more than 16 M residues (MC_TASK_W3)"; return -1; }
    if (dalloc(&h->d_res_base, H.res.size() + 128) || dalloc(&h->d_off, H.off.size()) || dalloc(&h->d_bstart, H.bstart.size()) || dalloc(&h->d_post, H.post.size() + 1) ||
        dalloc(&h->d_keys, H.keys.size()) || dalloc(&h->d_fam, (size_t)nseq) || dalloc(&h->d_T, 1) || dalloc(&h->d_P, 1)) return -1;
    HIPCK(hipMemset(h->d_res_base, MC_INV, H.res.size() + 128));
    h->d_res = h->d_res_base + 64;                                 // (k_gapped_lds reads 16 bytes at a time around a flank's first residues)
    HIPCK(hipMemcpy(h->d_res, H.res.data(), H.res.size(), hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(h->d_off, H.off.data(), H.off.size() * 4, hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(h->d_bstart, H.bstart.data(), H.bstart.size() * 4, hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(h->d_post, H.post.data(), H.post.size() * 4, hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(h->d_keys, H.keys.data(), H.keys.size() * 2, hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(h->d_fam, h->fam.data(), (size_t)nseq * 4, hipMemcpyHostToDevice));
    if (dalloc(&h->d_bitmap, H.bitmap.size())) return -1;
    HIPCK(hipMemcpy(h->d_bitmap, H.bitmap.data(), H.bitmap.size() * 4, hipMemcpyHostToDevice));
    if (dalloc(&h->d_filt, H.filt.size()) || dalloc(&h->d_wild, H.wild.size()) || dalloc(&h->d_pair, H.pair.size())) return -1;
    HIPCK(hipMemcpy(h->d_pair, H.pair.data(), H.pair.size() * 4, hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(h->d_wild, H.wild.data(), H.wild.size() * 4, hipMemcpyHostToDevice));
    if (dalloc(&h->d_rt, H.rt.size())) return -1;
    HIPCK(hipMemcpy(h->d_rt, H.rt.data(), H.rt.size() * 8, hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(h->d_filt, H.filt.data(), H.filt.size() * 4, hipMemcpyHostToDevice));
    if (!H.rec.empty()) {
        if (dalloc(&h->d_rec, H.rec.size())) return -1;
        HIPCK(hipMemcpy(h->d_rec, H.rec.data(), H.rec.size() * sizeof(McBucketRec), hipMemcpyHostToDevice));
    }
    MC_OT("  index upload", t0);
    if (const char *e = getenv("MC_PARTS")) { const int v = atoi(e); if (v >= 1) h->parts = v > MC_NCTX ? MC_NCTX : v; }   // (experiments)
    if (H.max_bucket > 2047) { g_err = "a seed bucket holds more than 2047 postings: the hit-order key cannot index it"; return -1; }
    // the position-parallel seed kernel is exact only when the frequency threshold is 0 and no letter frequency is 0
    h->fast_enum = (H.freq_thr == 0) && !H.rec.empty() && !getenv("MC_FORCE_SEQUENTIAL_ENUM");
    for (int g = 0; g < 10; g++) if (!(H.letter_p[g] > 0.0)) h->fast_enum = false;
    return 0;
}

// The directory mc_open() keeps built indexes in (mc_set_index_cache; empty: none).  Process-wide, set before the engines are opened.
static std::mutex g_ixc_mu;
static std::string g_ixc_dir;
extern "C" int mc_set_index_cache(const char *dir)
{
    std::unique_lock<std::mutex> lk(g_ixc_mu);
    g_ixc_dir = dir ? dir : "";
    return 0;
}

extern "C" mc_handle *mc_open(const char *const *names, const char *const *seqs, int32_t nseq, const int32_t *marker_family, int32_t nfam, int32_t device)
{
    mc_handle *h = new mc_handle();
    std::string err;
    double t0 = mc_now();
    std::string cache;
    uint64_t ih = 0;
    { std::unique_lock<std::mutex> lk(g_ixc_mu); cache = g_ixc_dir; }
    bool loaded = false;
    if (!cache.empty() && nseq > 0) {
        ih = mc_ixc_input_hash(names, seqs, nseq);
        char nm[64]; snprintf(nm, sizeof nm, "/index_%016llx.mcix", (unsigned long long)ih);
        cache += nm;
        loaded = mc_index_load(h->H, ih, nseq, cache.c_str());
        MC_OT(loaded ? "index cache: loaded" : "index cache: none / not usable", t0);
    }
    if (!loaded) {
        if (!mc_build_index(h->H, names, seqs, nseq, err)) { delete h; g_err = err; return nullptr; }
        MC_OT("mc_build_index", t0);
        if (!cache.empty()) { (void)mc_index_save(h->H, ih, cache.c_str()); MC_OT("index cache: written", t0); }
    }
    if (open_impl(h, marker_family, nfam, device) != 0) { std::string e = g_err; mc_close(h); g_err = e; return nullptr; }
    MC_OT("open_impl (device side)", t0);
    return h;
}

extern "C" mc_handle *mc_open_rapdb(const char *rapdb_path, int32_t device)
{
    mc_handle *h = new mc_handle();
    std::string err;
    if (!mc_load_rapdb(h->H, rapdb_path, err)) { delete h; g_err = err; return nullptr; }
    if (open_impl(h, nullptr, 1, device) != 0) { std::string e = g_err; mc_close(h); g_err = e; return nullptr; }
    return h;
}

extern "C" int32_t mc_marker_count(const mc_handle *h) { return h ? h->H.nseq : -1; }
extern "C" const char *mc_marker_name(const mc_handle *h, int32_t i) { return (h && i >= 0 && i < h->H.nseq) ? h->H.names[(size_t)i].c_str() : nullptr; }

extern "C" int mc_set_families(mc_handle *h, const int32_t *marker_family, int32_t nfam)
{
    if (!h || !marker_family) { g_err = "null argument"; return -1; }
    if (nfam < 1 || nfam > 32) { g_err = "1..32 gene families are supported"; return -1; }
    for (int i = 0; i < h->H.nseq; i++) if (marker_family[i] < 0 || marker_family[i] >= nfam) { g_err = "family index out of range"; return -1; }
    HIPCK(hipSetDevice(h->device));
    h->fam.assign(marker_family, marker_family + h->H.nseq);
    h->nfam = nfam;
    HIPCK(hipMemcpy(h->d_fam, h->fam.data(), (size_t)h->H.nseq * 4, hipMemcpyHostToDevice));
    h->run_set = false;                                            // per-family parameters have to be set again
    return 0;
}

// Host only (no GPU): `prerapsearch -d <fasta> -n <path>` - builds the index from the sequences and writes <path> and
// <path>.info in RAPSearch2 2.15's on-disk format.
extern "C" int mc_rapdb_write(const char *const *names, const char *const *seqs, int32_t nseq, const char *path)
{
    McHostIndex A;
    std::string err;
    if (!mc_build_index(A, names, seqs, nseq, err) || !mc_write_rapdb(A, path, err)) { g_err = err; return -1; }
    return 0;
}

// Host only (no GPU): is the database prerapsearch wrote the same index mc_open() builds from these sequences?
// 0 = identical (residues, offsets, buckets, postings in order, suffix keys); > 0 = number of the first differing part.
extern "C" int mc_rapdb_verify(const char *rapdb_path, const char *const *names, const char *const *seqs, int32_t nseq)
{
    McHostIndex A, B;
    std::string err;
    if (!mc_load_rapdb(A, rapdb_path, err) || !mc_build_index(B, names, seqs, nseq, err)) { g_err = err; return -1; }
    if (A.nseq != B.nseq || A.off != B.off) { g_err = "sequence count / offsets differ"; return 1; }
    if (A.res != B.res) { g_err = "residues differ"; return 2; }
    if (A.bstart != B.bstart) { g_err = "bucket sizes differ"; return 3; }
    if (A.post != B.post) { g_err = "posting order differs"; return 4; }
    if (A.keys != B.keys) { g_err = "suffix keys differ"; return 5; }
    if (A.names != B.names) { g_err = "names differ"; return 6; }
    return 0;
}

extern "C" int mc_index_view(const mc_handle *h, const uint8_t **res_codes, const uint32_t **offsets, const uint32_t **bucket_starts, const uint32_t **postings,
                             const uint16_t **keys, int64_t *nres, int64_t *npostings, uint32_t *freq_thr, double letter_p[10])
{
    if (!h) { g_err = "null handle"; return -1; }
    *res_codes = h->H.res_code.data(); *offsets = h->H.off.data(); *bucket_starts = h->H.bstart.data(); *postings = h->H.post.data(); *keys = h->H.keys.data();
    *nres = h->H.nres; *npostings = (int64_t)h->H.post.size(); *freq_thr = h->H.freq_thr;
    for (int i = 0; i < 10; i++) letter_p[i] = h->H.letter_p[i];
    return 0;
}

extern "C" int mc_set_run(mc_handle *h, int32_t read_len, double loge_thr, const double *min_cov, const double *min_score, const int32_t *max_aaid, const int32_t *aln_stat)
{
    if (!h) { g_err = "null handle"; return -1; }
    if (read_len < 18 || read_len > 3 * MC_MAXAA) { g_err = "read_len out of range (18..510)"; return -1; }
    HIPCK(hipSetDevice(h->device));
    double t0 = mc_now();
    mc_fill_tables(h->hT, h->H, read_len, loge_thr);
    MC_OT("set_run: tables", t0);
    if (mc_seg_fx_verify(h->hT, nullptr) != 0) { g_err = "internal: the fixed-point SEG tests disagree with the reference arithmetic"; return -1; }
    MC_OT("set_run: seg_fx_verify", t0);
    memset(&h->hP, 0, sizeof h->hP);
    h->hP.nfam = h->nfam; h->hP.read_len = read_len;
    for (int f = 0; f < h->nfam; f++) { h->hP.min_cov[f] = min_cov[f]; h->hP.min_score[f] = min_score[f]; h->hP.max_aaid[f] = max_aaid[f]; h->hP.aln_stat[f] = aln_stat[f]; }
    HIPCK(hipMemcpy(h->d_T, &h->hT, sizeof(McTables), hipMemcpyHostToDevice));
    if (!h->d_segtab) {   // Seg::getprob of every short window, tabulated once (ln n! does not depend on the run)
        std::vector<uint64_t> tab;
        mc_build_segtab(h->hT.lnfac, tab);
        if (dalloc(&h->d_segtab, tab.size())) return -1;
        HIPCK(hipMemcpy(h->d_segtab, tab.data(), tab.size() * 8, hipMemcpyHostToDevice));
    }
    HIPCK(hipMemcpy(h->d_P, &h->hP, sizeof(McClassPars), hipMemcpyHostToDevice));
    const int newFP = ((read_len / 3 + 2) + 3) & ~3;
    if (newFP != h->FP || read_len != h->read_len) for (McCtx &c : h->ctx) c.cap_reads = 0;   // pools are sized by read length and frame pitch
    h->read_len = read_len; h->FP = newFP; h->run_set = true;
    MC_OT("set_run: segtab, uploads", t0);
    return 0;
}

static int ensure_capacity(mc_handle *h, McCtx &c, int64_t nreads)
{
    if (nreads <= c.cap_reads) return 0;
    double t0 = mc_now();
    int64_t cap = nreads;
    if (cap > (1 << 21) - 1) { g_err = "batch larger than 2097151 reads"; return -1; }
    // pool sizes: generous multiples of what shotgun reads produce (75 seed hits, 23 kept HSPs, 5 gapped extensions per 150 bp
    // read of a real genome), scaled with the read length; a batch that still overflows is split by mc_search
    const int64_t L = h->read_len;
    c.cap_reads = 0;                                                // pools are being replaced: nothing is usable until all of them exist
    c.cap_tasks = (uint32_t)std::min<int64_t>(cap * (L + 32) + (1 << 20) + (int64_t)256 * 32 * MC_EN_BLK, 0x7fffffff);
    const int64_t ev_pad = (int64_t)256 * 8 * (MC_EV_BS / 64) * MC_EV_BLK;   // k_eval_seeds hands both pools out in blocks of MC_EV_BLK slots per wave: room for every wave's partly used last block
    c.cap_gaps = (uint32_t)std::min<int64_t>(cap * (L / 8 + 8) + (1 << 18) + ev_pad, (1 << 27) - 2);   // (k_gap_dedupe keeps task index + 1 in 27 bits of a table entry: more tasks than that overflow the pool and the range is split)
    // (round 4: HSPs L / 3 + 8 per read - 58 at 150 bp, where shotgun reads make 23 - instead of L / 2 + 16, rows 16 per read instead of 48:
    // allocating the pools of a 1 M-read batch took 0.8 s, most of the wall time of the reference's default run; a denser batch is split)
    c.cap_hsps = (uint32_t)std::min<int64_t>(cap * (L / 3 + 8) + (1 << 20) + ev_pad, 0x7fffffff);
    c.cap_rows = (uint32_t)std::min<int64_t>(cap * 16 + (1 << 20), 0x7fffffff);
    c.gap_threads_full = 16 * 1024;                                 // full-size DP rows for the last-resort launch (460 MB)
    if (dalloc(&c.d_frames_base, (size_t)cap * 6 * h->FP + 128) || dalloc(&c.d_tasks, c.cap_tasks) ||
        dalloc(&c.d_gaps, c.cap_gaps) || dalloc(&c.d_hsps, c.cap_hsps) || dalloc(&c.d_v, c.cap_hsps) ||
        dalloc(&c.d_tmp, (size_t)c.cap_hsps * 2) || dalloc(&c.d_k64, c.cap_hsps) || dalloc(&c.d_hkeys, c.cap_hsps) || dalloc(&c.d_hplace, c.cap_hsps) || dalloc(&c.d_places, c.cap_hsps) || dalloc(&c.d_idx, c.cap_hsps) ||
        dalloc(&c.d_idxo, c.cap_hsps) || dalloc(&c.d_heads, (size_t)cap + 2) || dalloc(&c.d_scan, (size_t)4100) || dalloc(&c.d_gsz, c.cap_hsps) || dalloc(&c.d_nv, (size_t)cap + 1) || dalloc(&c.d_rows, c.cap_rows) ||
        dalloc(&c.d_low, (size_t)cap + 64) || dalloc(&c.d_cand, (size_t)cap + 64) || dalloc(&c.d_nrow, (size_t)cap + 1) || dalloc(&c.d_rowoff, (size_t)cap + 1) || dalloc(&c.d_best, (size_t)cap + 1) || dalloc(&c.d_bestof, (size_t)cap + 1) ||
        dalloc(&c.d_gws_full, (size_t)c.gap_threads_full * MC_GAP_W) || dalloc(&c.d_retry, (size_t)c.cap_gaps * 2 + (size_t)cap + 1) || dalloc(&c.d_retry2, (size_t)c.cap_gaps * 2) || dalloc(&c.d_gleader, (size_t)c.cap_gaps) ||
        dalloc(&c.d_fout, (size_t)c.cap_gaps * 2))
        return -1;
    c.d_frames = c.d_frames_base + 64;
    HIPCK(hipMemsetAsync(c.d_frames_base, MC_INV, 64, c.stream));
    if (c.h_best) { (void)hipHostFree(c.h_best); c.h_best = nullptr; }
    HIPCK(hipHostMalloc((void **)&c.h_best, sizeof(McBestHit) * ((size_t)cap + 1), hipHostMallocDefault));
    c.h_best_cap = (size_t)cap + 1;
    size_t bytes = 0;                                               // (the flank list of the gapped stage is ordered by a 10-bit radix sort)
    HIPCK(rocprim::radix_sort_pairs_desc(nullptr, bytes, c.d_idx, c.d_idxo, c.d_idx, c.d_idxo, (size_t)c.cap_gaps * 2, 0, 10, c.stream));
    if (c.d_sorttmp) { (void)hipFree(c.d_sorttmp); c.d_sorttmp = nullptr; }
    HIPCK(hipMalloc(&c.d_sorttmp, bytes + 16));
    c.sorttmp_bytes = bytes;
    c.cap_reads = cap;
    HIPCK(hipStreamSynchronize(c.stream));
    MC_OT("ensure_capacity (pools)", t0);
    return 0;
}

extern "C" int mc_upload(mc_handle *h, const uint8_t *reads, int64_t nreads)
{
    if (!h || !h->run_set) { g_err = "mc_set_run() must be called first"; return -1; }
    HIPCK(hipSetDevice(h->device));
    const int64_t need = nreads * (int64_t)h->read_len + 16;       // capacity in bytes: the read length may change between runs
    if (need > h->cap_own) { if (dalloc(&h->d_reads, (size_t)need)) return -1; h->cap_own = need; }
    if (nreads) HIPCK(hipMemcpyAsync(h->d_reads, reads, (size_t)nreads * h->read_len, hipMemcpyHostToDevice, h->ctx[0].stream));
    HIPCK(hipStreamSynchronize(h->ctx[0].stream));
    h->reads_dev = h->d_reads; h->nreads = nreads;
    return 0;
}

extern "C" int mc_attach(mc_handle *h, const void *device_reads, int64_t nreads)
{
    if (!h || !h->run_set) { g_err = "mc_set_run() must be called first"; return -1; }
    h->reads_dev = (const uint8_t *)device_reads; h->nreads = nreads;
    return 0;
}

static float ev_ms(hipEvent_t a, hipEvent_t b) { float ms = 0; (void)hipEventElapsedTime(&ms, a, b); return ms; }

// The pipeline of one part, in five stages.  Each stage only ISSUES work on the part's stream and ends with an asynchronous copy
// of the device counters into pinned host memory; the next stage starts by waiting for that copy (stage_wait) and sizes its
// launches from it.  mc_run_range() interleaves the stages of its parts.
static int stage_wait(McCtx &c) { HIPCK(hipStreamSynchronize(c.stream)); return 0; }
static int counters_to_host(McCtx &c) { HIPCK(hipMemcpyAsync(c.h_c, c.d_counters, sizeof(uint32_t) * C_N, hipMemcpyDeviceToHost, c.stream)); return 0; }

// A: translation + SEG, seed enumeration, seed evaluation (gate, growth, ungapped X-drop)
static int stage_a(mc_handle *h, McCtx &c)
{
    const int64_t n = c.n;
    const int L = h->read_len, FP = h->FP;
    hipStream_t st = c.stream;
    McIndex X = dev_index(h);
    c.ntasks = c.ngaps = c.gpad = c.nh = c.nheads = c.nrows = c.nbest = c.nsegs = 0;
    HIPCK(hipMemsetAsync(c.d_counters, 0, sizeof(uint32_t) * C_N, st));
    HIPCK(hipMemsetAsync(c.d_stats, 0, sizeof(unsigned long long) * S_N, st));
    if (h->best_only) HIPCK(hipMemsetAsync(c.d_cand, 0, (size_t)n, st));
    HIPCK(hipMemsetAsync(c.d_low, 0, (size_t)n, st));
    HIPCK(hipEventRecord(c.ev[0], st));
    const int64_t threads = n * 6;
    const size_t lds_rest = (size_t)MC_TS_NLNF(FP) * 8 + (size_t)MC_TS_THREADS * MC_TS_STRIDE(FP);
    const size_t lds_staged = (size_t)MC_TS_STAGE(L) + lds_rest, lds_direct = (size_t)MC_TS_STAGE(0) + lds_rest;
    static const int ts_force = getenv("MC_TS_STAGED") ? atoi(getenv("MC_TS_STAGED")) : -1;
    const size_t cu_lds = 160 * 1024 - 1024;                      // (static LDS of the kernel and allocation granules)
    const bool staged = ts_force >= 0 ? ts_force != 0 : cu_lds / lds_staged >= cu_lds / lds_direct;   // staging stays while it does not cost a resident workgroup
    const size_t lds = staged ? lds_staged : lds_direct;
    if (staged) {
        if (lds > 48 * 1024) HIPCK(hipFuncSetAttribute((const void *)k_translate_seg<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        k_translate_seg<true><<<dim3((unsigned)((n + MC_TS_READS - 1) / MC_TS_READS)), dim3(MC_TS_THREADS), lds, st>>>(h->d_T, c.reads, L, n, c.d_frames, FP, h->d_segtab);
    } else {
        if (lds > 48 * 1024) HIPCK(hipFuncSetAttribute((const void *)k_translate_seg<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        k_translate_seg<false><<<dim3((unsigned)((n + MC_TS_READS - 1) / MC_TS_READS)), dim3(MC_TS_THREADS), lds, st>>>(h->d_T, c.reads, L, n, c.d_frames, FP, h->d_segtab);
    }
    HIPCK(hipEventRecord(c.ev[1], st));
#ifdef MC_EXP_TIMING
    {
        HIPCK(hipStreamSynchronize(st));
        unsigned long long acc[12], cnt[12];
        HIPCK(hipMemcpyFromSymbol(acc, HIP_SYMBOL(g_ts_acc), sizeof acc)); HIPCK(hipMemcpyFromSymbol(cnt, HIP_SYMBOL(g_ts_cnt), sizeof cnt));
        const char *nm[12] = {"flags", "advance", "numbering", "class-0 rounds", "class-1 rounds", "reduction", "owners", "mask", "staging", "translation", "write-out", ""};
        const double waves = (double)((n + MC_TS_READS - 1) / MC_TS_READS) * MC_TS_WAVES;
        for (int k = 0; k < 11; k++) fprintf(stderr, "ts-timing %-15s %9.1f cycles/wave  %8.2f entries/wave  total %8.1f Mcycles\n", nm[k], (double)acc[k] / waves, (double)cnt[k] / waves, acc[k] / 1e6);
        unsigned long long z[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        HIPCK(hipMemcpyToSymbol(HIP_SYMBOL(g_ts_acc), z, sizeof z)); HIPCK(hipMemcpyToSymbol(HIP_SYMBOL(g_ts_cnt), z, sizeof z));
    }
#endif
    if (h->fast_enum) {
        const size_t per_wave = MC_EN_WAVE_BYTES(h->count_traffic) + MC_EN_WAVE_LDS(FP, L);
        // Launch shape: the kernel needs 79 VGPRs (6 waves per SIMD) and is bound by instruction issue with some latency left to
        // hide - measured per 1 M reads of 150 bp: 16 waves per CU 6.77 ms, 20: 6.45, 24 (2 x 12, 3 x 8, 6 x 4 alike): 6.39.
        // So: as many waves per CU as the LDS holds, up to 24, in workgroups of 12 / 8 / 4 / 16 waves.
        int waves = 0, bpc = 1;
        {
            static const int shapes[][2] = {{12, 2}, {8, 3}, {4, 6}, {4, 5}, {16, 1}, {8, 2}, {4, 4}, {12, 1}, {4, 3}, {8, 1}, {4, 2}, {4, 1}};
            for (const auto &sh : shapes) if (!waves && (size_t)sh[1] * (64 + sh[0] * per_wave) <= 160 * 1024) { waves = sh[0]; bpc = sh[1]; }
        }
        if (!waves) { g_err = "reads too long for the seed kernel's LDS layout"; return -1; }
        if (const char *e = getenv("MC_EN_SHAPE")) { int a = 0, b = 0; if (sscanf(e, "%d,%d", &a, &b) == 2 && (a == 16 || a == 12 || a == 8 || a == 4) && b >= 1 && (size_t)b * (64 + a * per_wave) <= 160 * 1024) { waves = a; bpc = b; } }   // (experiments)
        const size_t lds2 = 64 + waves * per_wave;
        const int blocks = (int)std::min<int64_t>((int64_t)256 * bpc, (n + waves - 1) / waves);
#define MC_LAUNCH_EN(WV, CNT)                                                                                                                      \
    do {                                                                                                                                           \
        HIPCK(hipFuncSetAttribute((const void *)k_enumerate_t0<WV, CNT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));                  \
        k_enumerate_t0<WV, CNT><<<dim3(blocks), dim3(64 * WV), lds2, st>>>(h->d_T, X, h->d_bitmap, c.d_frames, FP, L, n, c.d_tasks, c.cap_tasks, \
                                                                            c.d_counters, c.d_stats);                                             \
    } while (0)
        if (h->count_traffic) { if (waves == 16) MC_LAUNCH_EN(16, true); else if (waves == 12) MC_LAUNCH_EN(12, true); else if (waves == 8) MC_LAUNCH_EN(8, true); else MC_LAUNCH_EN(4, true); }
        else { if (waves == 16) MC_LAUNCH_EN(16, false); else if (waves == 12) MC_LAUNCH_EN(12, false); else if (waves == 8) MC_LAUNCH_EN(8, false); else MC_LAUNCH_EN(4, false); }
#undef MC_LAUNCH_EN
    } else
        k_enumerate<<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st>>>(h->d_T, X, c.d_frames, FP, L, n, c.d_tasks, c.cap_tasks, c.d_counters, c.d_stats);
    HIPCK(hipEventRecord(c.ev[2], st));
    // the number of seed hits stays on the device: persistent workgroups walk the pool
    const size_t lds_ev = (size_t)(MC_EV_BS / 64) * MC_EV_QCAP * 32;   // a queue of survivors per wave: 32 KB per workgroup
    HIPCK(hipFuncSetAttribute((const void *)k_eval_seeds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_ev));
    static const unsigned ev_bpc = getenv("MC_EV_BPC") ? (unsigned)std::max(1, std::min(8, atoi(getenv("MC_EV_BPC")))) : (unsigned)MC_EV_BPC;   // (experiments)
    k_eval_seeds<<<dim3(256u * ev_bpc), dim3(MC_EV_BS), lds_ev, st>>>(h->d_T, X, c.d_frames, FP, L, c.d_tasks, c.d_counters + C_TASKS, c.cap_tasks, c.d_hsps, c.cap_hsps, c.d_gaps, c.cap_gaps, c.d_counters, h->d_P, h->d_fam, h->best_only ? c.d_cand : nullptr, c.d_hkeys, c.d_low, c.d_hplace);
    HIPCK(hipEventRecord(c.ev[3], st));
    return counters_to_host(c);
}

// B: gapped extension
static int stage_b(mc_handle *h, McCtx &c)
{
    const int L = h->read_len, FP = h->FP;
    hipStream_t st = c.stream;
    McIndex X = dev_index(h);
    if (c.h_c[C_OVERFLOW]) { g_err = "seed task / HSP / gap task buffer overflow"; return -2; }
    c.ntasks = c.h_c[C_TASKS];
    const uint32_t ngaps = c.ngaps = c.h_c[C_GAPS];                 // (slots of the pool: the padding of the waves' last blocks included)
    c.gpad = c.h_c[C_GPAD];
    if (ngaps) {
        // 1. group the tasks that extend the same ungapped segment and list the flanks of the distinct ones (k_gap_dedupe);
        // 2. order the flanks by DP size (the sort buffers of the HSP sort are idle at this point); 3. extend them with the DP rows
        // in LDS, those whose band leaves the window again with a wider one, the rest with full-size rows; 4. every task takes its
        // HSP from its group's flank results.  The counts of 2. - 4. stay on the device.
        static const int gap_refill = getenv("MC_GAP_REFILL") ? std::max(1, std::min(64, atoi(getenv("MC_GAP_REFILL")))) : MC_GAP_REFILL;   // (experiments)
        static const unsigned gap_wpc = getenv("MC_GAP_WPC") ? (unsigned)std::max(1, atoi(getenv("MC_GAP_WPC"))) : 8u;                      // waves per CU of the launch
        uint32_t slots = 1u << 16;
        while (slots < 2 * ngaps) slots <<= 1;
        if (slots > c.gtab_slots) { if (dalloc(&c.d_gtab, (size_t)slots)) return -1; c.gtab_slots = slots; }
        // The flank sort borrows the buffers of the HSP sort: 2 ngaps keys + 2 ngaps sorted keys in d_k64 (8 cap_hsps bytes), 2 ngaps
        // items in d_idx / d_idxo (4 cap_hsps bytes each).  The pools are sized so that ordinary batches fit (ensure_capacity); a batch
        // dense in gap tasks that does not is an overflow like any other: the range is run again in halves.
        if (2 * (uint64_t)ngaps > c.cap_hsps) { g_err = "gap task pool larger than the sort buffers"; return -2; }
        uint32_t *gk = (uint32_t *)c.d_k64, *gko = gk + 2 * (size_t)ngaps, *gi = c.d_idx, *gio = c.d_idxo;
        HIPCK(hipMemsetAsync(c.d_gtab, 0, (size_t)slots * 8, st));
        HIPCK(hipMemsetAsync(gk, 0, (size_t)ngaps * 8, st));
        k_gap_dedupe<<<dim3((ngaps + 255) / 256), dim3(256), 0, st>>>(X, L, c.d_gaps, ngaps, c.d_gtab, slots - 1, c.d_gleader, gk, gi, c.d_counters);
        size_t gbytes = c.sorttmp_bytes;
        HIPCK(rocprim::radix_sort_pairs_desc(c.d_sorttmp, gbytes, gk, gko, gi, gio, (size_t)ngaps * 2, 0, 10, st));
        k_gapped_lds<MC_GAP_WIN, 64><<<dim3(std::min<uint32_t>((2 * ngaps + 63) / 64, 256u * gap_wpc)), dim3(64), 0, st>>>(h->d_T, X, c.d_frames, FP, L, c.d_gaps, gio, c.d_counters + C_ITEMS, c.d_fout,
                                                                                                                 c.d_counters + C_RETRY, c.d_retry, gap_refill);
        k_gapped_lds<MC_GAP_WIN2, MC_GAP_LANES2><<<dim3(256u * 4u), dim3(64), 0, st>>>(h->d_T, X, c.d_frames, FP, L, c.d_gaps, c.d_retry, c.d_counters + C_RETRY, c.d_fout, c.d_counters + C_RETRY2, c.d_retry2, 1);
        k_gapped<<<dim3(c.gap_threads_full / 128), dim3(128), 0, st>>>(h->d_T, X, c.d_frames, FP, L, c.d_gaps, c.d_retry2, c.d_counters + C_RETRY2, c.d_fout, c.d_counters, c.d_gws_full, MC_GAP_W);
        k_gap_emit<<<dim3((ngaps + 255) / 256), dim3(256), 0, st>>>(h->d_T, X, L, c.d_gaps, ngaps, c.d_gleader, c.d_fout, c.d_hsps, c.cap_hsps, c.d_counters, h->d_P, h->d_fam, h->best_only ? c.d_cand : nullptr, c.d_hkeys, c.d_low, c.d_hplace);
    }
    HIPCK(hipEventRecord(c.ev[4], st));
    return counters_to_host(c);
}

// C: HSPs into per-read segments ordered by (subject, hit order); the reads that can print anything (see k_bin_count)
static int stage_c(mc_handle *h, McCtx &c)
{
    hipStream_t st = c.stream;
    if (c.h_c[C_OVERFLOW]) { g_err = "HSP buffer overflow"; return -2; }
    const uint32_t nslots = c.h_c[C_HSPS];                         // used slots of the pool, the padding of the waves' last blocks included
    c.nh_all = nslots - c.h_c[C_HPAD];
    c.nh = c.nh_all;                                               // (best hits only: the reads that can be classified are selected on the device - cand)
    if (c.nh) {
        const uint32_t n = (uint32_t)c.n;
        const uint8_t *cand = h->best_only ? c.d_cand : nullptr;
        uint32_t *cur = c.d_heads + 1;                             // heads[0] = 0; cur[r]: counts -> starts -> ends = heads[r + 1]
        uint64_t *keys = c.d_k64;                                  // (the gapped stage's sort buffers: free again)
        uint32_t *slots = c.d_idxo, *order = c.d_idx, *heavy = c.d_retry2, *heavy2 = c.d_retry2 + c.cap_gaps;   // (lists of at most n reads: cap_gaps >= 10 n)
        static const unsigned bin_blocks = getenv("MC_BIN_BLOCKS") ? (unsigned)std::max(1, atoi(getenv("MC_BIN_BLOCKS"))) : 256u * 8u;   // (experiments)
        static const bool order_serial = getenv("MC_ORDER_SERIAL") != nullptr;                                                              // (experiments: the three order kernels one after the other)
        HIPCK(hipMemsetAsync(c.d_heads, 0, ((size_t)n + 2) * sizeof(uint32_t), st));
        HIPCK(hipMemsetAsync(c.d_nrow, 0, ((size_t)n + 1) * sizeof(uint32_t), st));
        HIPCK(hipMemsetAsync(order, 0xFF, (size_t)nslots * sizeof(uint32_t), st));
        k_bin_count<<<dim3(bin_blocks), dim3(256), 0, st>>>(c.d_hkeys, c.d_counters, c.cap_hsps, cand, cur);
        if (mc_scan_u32(cur, n, cur, c.d_scan, st)) return -1;
        k_bin_scatter<<<dim3(bin_blocks), dim3(256), 0, st>>>(c.d_hkeys, c.d_counters, c.cap_hsps, cand, cur, c.d_hplace, keys, c.d_places, slots);
        // the long segments beside the short ones (the few segments of more than 512 HSPs are a long tail on a nearly empty GPU)
        uint32_t *heavy3 = heavy2 + c.cap_gaps / 2;
        k_order_lists<<<dim3((n + 255) / 256), dim3(256), 0, st>>>(c.d_heads, n, c.d_counters, heavy, heavy2, heavy3);
        hipStream_t side = order_serial ? st : c.side;
        if (!order_serial) { HIPCK(hipEventRecord(c.ev_fork, st)); HIPCK(hipStreamWaitEvent(c.side, c.ev_fork, 0)); }
        HIPCK(hipFuncSetAttribute((const void *)k_order_heavy<1024, MC_ORDER_LDS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(MC_ORDER_LDS * 18)));
        k_order_heavy<1024, MC_ORDER_LDS><<<dim3(256u), dim3(1024), MC_ORDER_LDS * 18, side>>>(keys, c.d_places, slots, c.d_heads, heavy3, c.d_counters + C_ORDER3, c.d_counters + C_OTAKE3, c.d_low, order, c.d_gsz, c.d_nv, c.d_nrow, (uint64_t *)c.d_tmp);
        k_order_heavy<256, MC_ORDER_MID><<<dim3(256u * 4u), dim3(256), MC_ORDER_MID * 18, side>>>(keys, c.d_places, slots, c.d_heads, heavy2, c.d_counters + C_ORDER2, c.d_counters + C_OTAKE2, c.d_low, order, c.d_gsz, c.d_nv, c.d_nrow, (uint64_t *)c.d_tmp);
        if (!order_serial) HIPCK(hipEventRecord(c.ev_join, c.side));
        k_order_light<<<dim3((n + MC_OL_READS - 1) / MC_OL_READS), dim3(256), 0, st>>>(keys, c.d_places, slots, c.d_heads, n, c.d_low, order, c.d_gsz, c.d_nv, c.d_nrow);
        k_order_heavy<64, MC_ORDER_SMALL><<<dim3(256u * 16u), dim3(64), MC_ORDER_SMALL * 18, st>>>(keys, c.d_places, slots, c.d_heads, heavy, c.d_counters + C_ORDER, c.d_counters + C_OTAKE, c.d_low, order, c.d_gsz, c.d_nv, c.d_nrow, (uint64_t *)c.d_tmp);
        if (!order_serial) HIPCK(hipStreamWaitEvent(st, c.ev_join, 0));
        k_order_copy<<<dim3(256u * 8u), dim3(256), 0, st>>>(order, c.d_gsz, c.d_hsps, c.d_heads, n, c.d_v);
        if (getenv("MC_BIN_STATS")) {                                  // (development aid: the sizes of the segments)
            std::vector<uint32_t> hh((size_t)n + 1);
            HIPCK(hipStreamSynchronize(st));
            HIPCK(hipMemcpy(hh.data(), c.d_heads, ((size_t)n + 1) * 4, hipMemcpyDeviceToHost));
            const uint32_t lim[9] = {16, 32, 64, 128, 512, 2048, 8192, 32768, 0xFFFFFFFFu};
            uint64_t cnt[9] = {0}, sum[9] = {0};
            std::vector<uint32_t> top;
            for (uint32_t r = 0; r < n; r++) { const uint32_t k = hh[r + 1] - hh[r]; if (!k) continue; for (int b = 0; b < 9; b++) if (k <= lim[b]) { cnt[b]++; sum[b] += k; break; } if (k > 2048) top.push_back(k); }
            std::sort(top.rbegin(), top.rend());
            fprintf(stderr, "bin-stats segments <=16 32 64 128 512 2048 8192 32768 more: reads"); for (int b = 0; b < 9; b++) fprintf(stderr, " %llu", (unsigned long long)cnt[b]);
            fprintf(stderr, "; HSPs"); for (int b = 0; b < 9; b++) fprintf(stderr, " %llu", (unsigned long long)sum[b]);
            fprintf(stderr, "; largest:"); for (size_t i = 0; i < top.size() && i < 12; i++) fprintf(stderr, " %u", top[i]); fprintf(stderr, "\n");
        }
    }
    HIPCK(hipEventRecord(c.ev[5], st));
    return 0;
}

// D: per-read finishing (linking, ranking, cap, classification), rows into m8 order
static int stage_d(mc_handle *h, McCtx &c)
{
    hipStream_t st = c.stream;
    McIndex X = dev_index(h);
    const uint32_t nh = c.nh, nheads = c.nheads = nh ? (uint32_t)c.n : 0u;   // (every read has a segment, most of them empty or unmarked)
    if (nh) {
        // the thread-per-read kernel (reads with few HSPs) on this stream, the wave-per-read kernels one after the other on a
        // second one (each hands the reads its LDS arrays cannot hold to the next)
        uint32_t *d_heavy = c.d_retry, *d_heavy2 = c.d_retry + c.cap_gaps / 2, *d_heavy3 = c.d_retry + c.cap_gaps;      // (d_retry is free again: the gap tasks are done)
        uint32_t *d_light = c.d_retry + c.cap_gaps + c.cap_gaps / 2;
        const uint32_t light_pitch = (uint32_t)c.cap_reads + 1;
        k_heavy_lists<<<dim3((nheads + 255) / 256), dim3(256), 0, st>>>(c.d_nv, nheads, c.d_nrow, c.d_bestof, c.d_counters, d_heavy, d_light, light_pitch, h->best_only ? MC_FH_MIN_BEST : MC_FH_MIN);
        HIPCK(hipEventRecord(c.ev_fork, st));
        {
            const size_t l1 = (size_t)MC_FH_N1 * 16 + 3 * (size_t)(MC_FH_N1 + 2) * 2, l2 = (size_t)MC_FH_N2 * 16 + 3 * (size_t)(MC_FH_N2 + 2) * 2, l3 = (size_t)MC_FH_N3 * 16 + 3 * (size_t)(MC_FH_N3 + 2) * 2;
            HIPCK(hipFuncSetAttribute((const void *)k_finish_heavy<MC_FH_N3, C_HEAVY3, -1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l3));
            HIPCK(hipStreamWaitEvent(c.side, c.ev_fork, 0));
            k_finish_heavy<MC_FH_N1, C_HEAVY, C_HEAVY2><<<dim3(256 * 12), dim3(64), l1, c.side>>>(h->d_T, X, h->d_P, h->d_fam, c.d_nv, c.d_heads, nheads, c.d_v, c.d_tmp, c.first_read_id,
                                                                                                   c.d_nrow, c.d_bestof, c.d_counters, d_heavy, d_heavy, d_heavy2);
            k_finish_heavy<MC_FH_N2, C_HEAVY2, C_HEAVY3><<<dim3(256 * 3), dim3(64), l2, c.side>>>(h->d_T, X, h->d_P, h->d_fam, c.d_nv, c.d_heads, nheads, c.d_v, c.d_tmp, c.first_read_id,
                                                                                                   c.d_nrow, c.d_bestof, c.d_counters, d_heavy, d_heavy2, d_heavy3);
            k_finish_heavy<MC_FH_N3, C_HEAVY3, -1><<<dim3(256), dim3(64), l3, c.side>>>(h->d_T, X, h->d_P, h->d_fam, c.d_nv, c.d_heads, nheads, c.d_v, c.d_tmp, c.first_read_id,
                                                                                         c.d_nrow, c.d_bestof, c.d_counters, d_heavy, d_heavy3, nullptr);
            // MergeRes' heap sort of all of them (a lane per read), then their rows (a wave per read)
            const size_t lh = (size_t)(MC_MAX_M8 + 2) * 64 * 4;
            HIPCK(hipFuncSetAttribute((const void *)k_heap_lanes, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lh));
            k_heap_lanes<<<dim3(256), dim3(64), lh, c.side>>>(c.d_heads, nheads, nh, c.d_tmp, c.d_nrow, c.d_counters, d_heavy);
            k_heavy_rows<<<dim3(256 * 12), dim3(64), 0, c.side>>>(h->d_T, X, h->d_P, h->d_fam, c.d_heads, nheads, c.d_v, c.d_tmp, c.first_read_id, c.d_nrow, c.d_bestof, c.d_counters, d_heavy);
            HIPCK(hipEventRecord(c.ev_join, c.side));
        }
        // the light reads: the four size classes side by side (the counts stay on the device; blocks past a class' count leave at once)
        k_finish<<<dim3((nheads + 255) / 256, 4), dim3(256), 0, st>>>(h->d_T, X, h->d_P, h->d_fam, c.d_nv, c.d_heads, nheads, c.d_v, c.d_tmp,
                                                                       c.first_read_id, c.d_nrow, c.d_bestof, d_light, light_pitch, c.d_counters + C_LIGHT0);
        HIPCK(hipStreamWaitEvent(st, c.ev_join, 0));
        if (mc_scan_u32(c.d_nrow, nheads, c.d_rowoff, c.d_scan, st)) return -1;
        if (h->rows_ever) HIPCK(hipStreamWaitEvent(st, h->ev_rows, 0));   // (the rows of the run before may still be leaving d_rows)
        k_emit_rows<<<dim3((nheads + 255) / 256), dim3(256), 0, st>>>(c.d_heads, nheads, c.d_nrow, c.d_rowoff, c.d_tmp, c.d_rows, c.cap_rows, c.d_bestof, c.d_best, c.d_counters, h->best_only ? 0 : 1);
    }
    HIPCK(hipEventRecord(c.ev[6], st));
#ifdef MC_EXP_TIMING
    {
        HIPCK(hipStreamSynchronize(st)); HIPCK(hipStreamSynchronize(c.side));
        unsigned long long acc[8], cnt[8];
        HIPCK(hipMemcpyFromSymbol(acc, HIP_SYMBOL(g_fh_acc), sizeof acc)); HIPCK(hipMemcpyFromSymbol(cnt, HIP_SYMBOL(g_fh_cnt), sizeof cnt));
        {
            unsigned long long fr[8];
            HIPCK(hipMemcpyFromSymbol(fr, HIP_SYMBOL(g_fr_acc), sizeof fr));
            const char *fn[7] = {"groups", "items", "std::sort", "threshold, keys", "heap sort", "rows", "classification"};
            for (int k = 0; k < 7; k++) fprintf(stderr, "fr-timing %-17s total %9.1f Mcycles (thread wall time, summed)\n", fn[k], fr[k] / 1e6);
            unsigned long long z8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            HIPCK(hipMemcpyToSymbol(HIP_SYMBOL(g_fr_acc), z8, sizeof z8));
            HIPCK(hipMemcpyFromSymbol(fr, HIP_SYMBOL(g_fr_acc2), sizeof fr));
            const char *gn[8] = {"-", "sort by frame", "sort by start, stable", "choice", "sum statistics", "copies", "groups (count)", "groups linked (count)"};
            for (int k = 1; k < 8; k++) fprintf(stderr, "fg-timing %-22s %12.1f M\n", gn[k], fr[k] / 1e6);
            HIPCK(hipMemcpyToSymbol(HIP_SYMBOL(g_fr_acc2), z8, sizeof z8));
        }
        {
            unsigned long long ev[8];
            HIPCK(hipMemcpyFromSymbol(ev, HIP_SYMBOL(g_ev_acc), sizeof ev));
            const char *en[5] = {"barriers, flush", "record, first reads", "growth, gate, X-drop", "HSP", "staging"};
            for (int k = 0; k < 5; k++) fprintf(stderr, "ev-timing %-21s total %9.1f Mcycles (lane 0 of every wave)\n", en[k], ev[k] / 1e6);
            unsigned long long z8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            HIPCK(hipMemcpyToSymbol(HIP_SYMBOL(g_ev_acc), z8, sizeof z8));
        }
        const char *nm[8] = {"group starts", "groups", "scan, items", "sort", "threshold, ranks", "heap sort", "rows", "other"};
        for (int k = 0; k < 8; k++) fprintf(stderr, "fh-timing %-17s total %9.1f Mcycles %9llu entries\n", nm[k], acc[k] / 1e6, cnt[k]);
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        HIPCK(hipMemcpyToSymbol(HIP_SYMBOL(g_fh_acc), z, sizeof z)); HIPCK(hipMemcpyToSymbol(HIP_SYMBOL(g_fh_cnt), z, sizeof z));
    }
#endif
    HIPCK(hipMemcpyAsync(c.h_stats, c.d_stats, sizeof(unsigned long long) * S_N, hipMemcpyDeviceToHost, st));
    return counters_to_host(c);
}

// E: rows (final order and ABI layout: McRow == mc_row) and best hits into pinned host memory; rows_at = where this part's rows
// go in the handle's row buffer
static int stage_e(mc_handle *h, McCtx &c, size_t rows_at)
{
    hipStream_t st = c.stream;
    if (c.nrows) HIPCK(hipMemcpyAsync(h->pin_rows + rows_at, c.d_rows, sizeof(McRow) * c.nrows, hipMemcpyDeviceToHost, h->rows_stream));   // (the part's stream has been waited for: d_rows is final)
    if (c.nbest) HIPCK(hipMemcpyAsync(c.h_best, c.d_best, sizeof(McBestHit) * c.nbest, hipMemcpyDeviceToHost, st));
    return 0;
}

static void rows_wait(mc_handle *h)
{   // the rows of the last run are on their way to the host: wait for them
    if (h->rows_pending) { (void)hipEventSynchronize(h->ev_rows); h->rows_pending = false; }
}

static void stats_add(mc_stats &tot, const mc_stats &s)
{
    tot.reads += s.reads; tot.seed_tasks += s.seed_tasks; tot.gap_tasks += s.gap_tasks; tot.hsps += s.hsps; tot.rows += s.rows;
    tot.reads_with_rows += s.reads_with_rows; tot.classified += s.classified; tot.bucket_lookups += s.bucket_lookups; tot.key_probes += s.key_probes;
    tot.seed_exact_asks += s.seed_exact_asks; tot.seed_wild_asks += s.seed_wild_asks; tot.seed_pair_asks += s.seed_pair_asks; tot.seed_probes += s.seed_probes;
    tot.ms_translate += s.ms_translate; tot.ms_seed += s.ms_seed; tot.ms_eval += s.ms_eval; tot.ms_gapped += s.ms_gapped;
    tot.ms_sort += s.ms_sort; tot.ms_finish += s.ms_finish; tot.ms_total += s.ms_total;
    tot.range_splits += s.range_splits;
}

static int run_range_once(mc_handle *h, int64_t first, int64_t count, int64_t first_read_id);

// A range whose seed hits / HSPs / rows overflow the pools sized for ordinary shotgun reads (-2 from the pipeline) is run again in
// halves, and their results joined: the caller sees one range either way.
extern "C" int mc_run_range(mc_handle *h, int64_t first, int64_t count, int64_t first_read_id)
{
    int rc = run_range_once(h, first, count, first_read_id);
    if (rc != -2 || count <= 1) return rc;
    std::vector<mc_row> &rows = h->split_rows; rows.clear();
    std::vector<mc_best_hit> best; mc_stats tot; memset(&tot, 0, sizeof tot);
    tot.range_splits = 1;
    int64_t off = 0, step = std::max<int64_t>(1, count / 2);
    while (off < count) {
        const int64_t nb = std::min<int64_t>(step, count - off);
        rc = run_range_once(h, first + off, nb, first_read_id + off);
        if (rc == -2 && nb > 1) { step = std::max<int64_t>(1, nb / 2); tot.range_splits++; continue; }
        if (rc) return rc;
        rows_wait(h);
        rows.insert(rows.end(), h->res_rows, h->res_rows + h->n_res_rows);
        best.insert(best.end(), h->best.begin(), h->best.end());
        stats_add(tot, h->stats);
        off += nb;
    }
    h->res_rows = rows.data(); h->n_res_rows = (int64_t)rows.size(); h->best.swap(best); h->stats = tot;
    return 0;
}

static int run_range_once(mc_handle *h, int64_t first, int64_t count, int64_t first_read_id)
{
    if (!h || !h->run_set) { g_err = "mc_set_run() must be called first"; return -1; }
    if (first < 0 || count < 0 || first + count > h->nreads) { g_err = "range outside the resident read set"; return -1; }
    if (count > (1 << 21) - 1) { g_err = "batch larger than 2097151 reads"; return -1; }
    HIPCK(hipSetDevice(h->device));
    memset(&h->stats, 0, sizeof h->stats);
    h->res_rows = nullptr; h->n_res_rows = 0; h->best.clear();
    h->pin_cur ^= 1; h->pin_rows = h->pin_slot[h->pin_cur]; h->pin_cap = h->pin_slot_cap[h->pin_cur];   // (the other slot may still be receiving the rows of the run before)
    h->stats.reads = count;
    if (count == 0) return 0;
    // parts: two halves (small ranges: one part)
    const int np = (count >= 65536 && h->parts > 1) ? MC_NCTX : 1;
    int64_t at = 0;
    for (int p = 0; p < np; p++) {
        McCtx &c = h->ctx[p];
        const int64_t n = (count - at) / (np - p);
        if (ensure_capacity(h, c, n)) return -1;
        c.reads = h->reads_dev + (first + at) * h->read_len; c.n = n; c.first_read_id = first_read_id + at;
        at += n;
    }
    int rc = 0;
#define MC_ALL(stmt) for (int p = 0; p < np && rc == 0; p++) { McCtx &c = h->ctx[p]; stmt; }
    MC_ALL(rc = stage_a(h, c))
    MC_ALL(if ((rc = stage_wait(c)) == 0) rc = stage_b(h, c))
    MC_ALL(if ((rc = stage_wait(c)) == 0 && (rc = stage_c(h, c)) == 0) rc = stage_d(h, c))   // (C leaves its counts on the device: D is issued behind it)
    // E, part by part: as soon as a part's finishing is done its rows start travelling, while the next part is still finishing
    size_t nrows = 0;
    for (int p = 0; p < np && rc == 0; p++) {
        McCtx &c = h->ctx[p];
        if ((rc = stage_wait(c)) != 0) break;
        if (c.h_c[C_OVERFLOW]) { g_err = "row buffer overflow"; rc = -2; break; }
        c.nrows = (c.nh && !h->best_only) ? c.h_c[C_ROWS] : 0u; c.nsegs = c.h_c[C_SEGS]; c.nbest = c.h_c[C_BEST];
        const size_t need = nrows + c.nrows;
        if (need > h->pin_cap) {                                     // grow the pinned row buffer (with room for the parts still to come)
            (void)hipStreamSynchronize(h->rows_stream);                  // (copies of earlier parts / of the run before write into the old buffers)
            const size_t want = need + (size_t)(np - 1 - p) * ((size_t)c.nrows + c.nrows / 4) + need / 4 + 1024;
            mc_row *nb = nullptr;
            if (hipHostMalloc((void **)&nb, want * sizeof(mc_row), hipHostMallocDefault) != hipSuccess) { g_err = "out of pinned host memory for the rows"; rc = -1; break; }
            if (nrows) memcpy(nb, h->pin_rows, nrows * sizeof(mc_row));
            if (h->pin_rows) (void)hipHostFree(h->pin_rows);
            h->pin_rows = nb; h->pin_cap = want; h->pin_slot[h->pin_cur] = nb; h->pin_slot_cap[h->pin_cur] = want;
            const int other = h->pin_cur ^ 1;                            // the other slot grows with it (pinning 300 MB takes 40 ms: not in the middle of a later run)
            if (h->pin_slot_cap[other] < want) {
                mc_row *ob = nullptr;
                if (hipHostMalloc((void **)&ob, want * sizeof(mc_row), hipHostMallocDefault) == hipSuccess) {
                    if (h->pin_slot[other]) (void)hipHostFree(h->pin_slot[other]);
                    h->pin_slot[other] = ob; h->pin_slot_cap[other] = want;
                }
            }
        }
        rc = stage_e(h, c, nrows);
        nrows += c.nrows;
    }
    if (rc) { for (int p = 0; p < np; p++) (void)hipStreamSynchronize(h->ctx[p].stream); (void)hipStreamSynchronize(h->rows_stream); return rc; }
    if (nrows) { HIPCK(hipEventRecord(h->ev_rows, h->rows_stream)); h->rows_pending = true; h->rows_ever = true; }
    MC_ALL(rc = stage_wait(c))
#undef MC_ALL
    if (rc) return rc;
    h->res_rows = h->pin_rows; h->n_res_rows = (int64_t)nrows;
    // classify_reads meets the reads in input order
    for (int p = 0; p < np; p++) {
        McCtx &c = h->ctx[p];
        std::sort(c.h_best, c.h_best + c.nbest, [](const McBestHit &x, const McBestHit &y) { return x.read < y.read; });
        for (uint32_t i = 0; i < c.nbest; i++) { const McBestHit &x = c.h_best[i]; mc_best_hit o; o.read = x.read; o.family = x.family; o.aln = x.aln; o.target_len = x.target_len; o.bits = x.bits; h->best.push_back(o); }
#ifdef MC_EXP_TIMING
        { const char *nm[6] = {"staging/other", "append", "lookup", "push", "setup", "expand"}; for (int k = 0; k < 6; k++) fprintf(stderr, "timing %-14s %8.3f Mcycles/wave-avg  %10llu entries\n", nm[k], (double)c.h_stats[4 + k] / 4096.0 / 1e6, c.h_stats[10 + k]); }
#endif
        h->stats.bucket_lookups += (int64_t)c.h_stats[S_LOOKUPS]; h->stats.key_probes += (int64_t)c.h_stats[S_KEYPROBES]; h->stats.seed_tasks += (int64_t)c.h_stats[S_TASKS];
        h->stats.seed_exact_asks += (int64_t)c.h_stats[S_EXACT]; h->stats.seed_wild_asks += (int64_t)c.h_stats[S_WILD]; h->stats.seed_pair_asks += (int64_t)c.h_stats[S_PAIRS]; h->stats.seed_probes += (int64_t)c.h_stats[S_PROBES];
        h->stats.gap_tasks += c.ngaps - c.gpad; h->stats.hsps += c.nh_all; h->stats.rows += c.nrows; h->stats.reads_with_rows += c.nsegs;
        // kernel times: HIP events on the part's own stream (the parts overlap, so the sums exceed the wall time of the call)
        h->stats.ms_translate += ev_ms(c.ev[0], c.ev[1]); h->stats.ms_seed += ev_ms(c.ev[1], c.ev[2]); h->stats.ms_eval += ev_ms(c.ev[2], c.ev[3]);
        h->stats.ms_gapped += ev_ms(c.ev[3], c.ev[4]); h->stats.ms_sort += ev_ms(c.ev[4], c.ev[5]); h->stats.ms_finish += ev_ms(c.ev[5], c.ev[6]); h->stats.ms_total += ev_ms(c.ev[0], c.ev[6]);
    }
    h->stats.classified = (int64_t)h->best.size();
    return 0;
}

extern "C" int mc_set_counting(mc_handle *h, int on)
{
    if (!h) { g_err = "null handle"; return -1; }
    h->count_traffic = on != 0;
    return 0;
}

extern "C" int mc_set_parts(mc_handle *h, int parts)
{
    if (!h || parts < 1) { g_err = "bad argument"; return -1; }
    h->parts = parts > MC_NCTX ? MC_NCTX : parts;
    return 0;
}

extern "C" int mc_run(mc_handle *h, int64_t first_read_id) { return h ? mc_run_range(h, 0, h->nreads, first_read_id) : -1; }

// The streaming form of the pipeline: batches of reads are fetched from a host-side source into pinned staging memory and
// uploaded by a thread of their own (two staging / device buffers in turn) while the calling thread runs mc_run_range() on the
// batch before - upload and search overlap.
#define MC_STREAM_BATCH 2000000
static int64_t stream_batch()
{   // reads per batch of the streaming pipeline (MC_STREAM_BATCH in the environment: tests deal small batches)
    if (const char *e = getenv("MC_STREAM_BATCH")) { const long long v = atoll(e); if (v >= 1000 && v <= MC_STREAM_BATCH) return (int64_t)v; }
    return MC_STREAM_BATCH;
}
struct McBatchSlot { uint8_t *pin = nullptr, *dev = nullptr; int64_t n = 0, first = 0; int state = 0; /* 0 free, 1 ready, 2 end / error */ int64_t rc = 0; };

// fetch(dst, max, &first) copies the next batch of at most `max` reads into dst, stores the index of its first read and returns
// how many there were (0: the source has ended; < 0: its error).
static int run_stream(mc_handle *h, const std::function<int64_t(uint8_t *, int64_t, int64_t *)> &fetch, int64_t first_read_id, int64_t expect_reads = 0)
{
    HIPCK(hipSetDevice(h->device));
    const int64_t BMAX = MC_STREAM_BATCH, B = stream_batch(), L = h->read_len;
    double t0 = mc_now();
    // The largest batch of this run: a quarter of the reads the caller expects (a power of two between 256 k and 2 M; 2 M when it
    // does not know).  Staging buffers and pools are sized for it ONCE, before the first batch - pinning 2 x 300 MB and allocating
    // (then re-allocating, as the batches grew) the pools of ever larger batches was 1 s of the 1.2 - 2 s of the reference's
    // default run, one run_pipeline of 1 - 2 M reads per process.
    int64_t bmax_run = B;
    if (B == BMAX && expect_reads > 0) { bmax_run = 262144; while (bmax_run < BMAX && bmax_run * 4 < expect_reads) bmax_run <<= 1; bmax_run = std::min(bmax_run, BMAX); }
    const size_t stage_bytes = (size_t)(bmax_run * L + 64);
    if (!h->stage_pin[0] || h->stage_bytes < stage_bytes) {
        for (int k = 0; k < 2; k++) {
            if (h->stage_pin[k]) { (void)hipHostFree(h->stage_pin[k]); h->stage_pin[k] = nullptr; }
            if (h->stage_dev[k]) { (void)hipFree(h->stage_dev[k]); h->stage_dev[k] = nullptr; }
            HIPCK(hipHostMalloc((void **)&h->stage_pin[k], stage_bytes, hipHostMallocDefault));
            HIPCK(hipMalloc((void **)&h->stage_dev[k], stage_bytes));
        }
        h->stage_bytes = stage_bytes;
        MC_OT("run_stream: staging buffers", t0);
    }
    if (!h->copy_stream) HIPCK(hipStreamCreate(&h->copy_stream));
    if (expect_reads > 0 && ensure_capacity(h, h->ctx[0], std::min<int64_t>(bmax_run, expect_reads))) return -1;
    McBatchSlot slot[2];
    for (int k = 0; k < 2; k++) { slot[k].pin = h->stage_pin[k]; slot[k].dev = h->stage_dev[k]; }
    std::mutex mu; std::condition_variable cv;
    bool abort_up = false;
    std::string up_err;
    std::thread uploader([&] {
        (void)hipSetDevice(h->device);
        int nb = 0;
        for (int k = 0;; k ^= 1) {
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return slot[k].state == 0 || abort_up; }); if (abort_up) return; }
            int64_t at = 0;
            // The first batches are small - 256 k, 512 k, 1 M reads, then 2 M: the device starts after 4 ms of parsing instead of 33,
            // which is a third of the wall time of the default run (2 M sampled reads); later batches have the full size, where the
            // fixed cost of a range (~1 ms) no longer shows.
            const int64_t want = B == BMAX ? std::min<int64_t>(bmax_run, (int64_t)262144 << std::min(nb, 3)) : B;
            nb++;
            const int64_t n = fetch(slot[k].pin, want, &at);
            int64_t rc = n;
            if (n > 0) {
                hipError_t e = hipMemcpyAsync(slot[k].dev, slot[k].pin, (size_t)(n * L), hipMemcpyHostToDevice, h->copy_stream);
                if (e == hipSuccess) e = hipStreamSynchronize(h->copy_stream);
                if (e != hipSuccess) { up_err = std::string("upload: ") + hipGetErrorString(e); rc = -1; }
            }
            std::unique_lock<std::mutex> lk(mu);
            slot[k].n = n > 0 ? n : 0; slot[k].first = at; slot[k].rc = rc; slot[k].state = rc > 0 ? 1 : 2;
            cv.notify_all();
            if (rc <= 0) return;
        }
    });
    std::vector<mc_row> &all_rows = h->all_rows; all_rows.clear();
    if (h->keep_rows && expect_reads > 0) all_rows.reserve((size_t)expect_reads * 2 + 1024);   // (shotgun reads of real genomes: 1.9 rows per read; untouched pages cost nothing)
    std::vector<mc_best_hit> all_best; mc_stats tot; memset(&tot, 0, sizeof tot);
    int rc = 0;
    const uint8_t *saved_reads = h->reads_dev; const int64_t saved_n = h->nreads;
    for (int k = 0;; k ^= 1) {
        { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return slot[k].state != 0; }); }
        if (slot[k].state == 2) { if (slot[k].rc < 0) { rc = (int)slot[k].rc; if (!up_err.empty()) g_err = up_err; } break; }
        h->reads_dev = slot[k].dev; h->nreads = slot[k].n;
        rc = mc_run_range(h, 0, slot[k].n, first_read_id + slot[k].first);      // (a pool overflow is answered inside: smaller ranges)
        if (rc == 0) {
            if (h->keep_rows) { rows_wait(h); all_rows.insert(all_rows.end(), h->res_rows, h->res_rows + h->n_res_rows); }
            all_best.insert(all_best.end(), h->best.begin(), h->best.end());
            stats_add(tot, h->stats);
        }
        if (rc) break;
        { std::unique_lock<std::mutex> lk(mu); slot[k].state = 0; cv.notify_all(); }
    }
    { std::unique_lock<std::mutex> lk(mu); abort_up = true; cv.notify_all(); }
    uploader.join();
    h->reads_dev = saved_reads; h->nreads = saved_n;
    if (rc) return rc;
    h->res_rows = all_rows.data(); h->n_res_rows = (int64_t)all_rows.size(); h->best.swap(all_best); h->stats = tot;
    return 0;
}

extern "C" int mc_search(mc_handle *h, const uint8_t *reads, int64_t nreads, int64_t first_read_id)
{
    if (!h || !h->run_set) { g_err = "mc_set_run() must be called first"; return -1; }
    if (nreads < 0 || (nreads > 0 && !reads)) { g_err = "bad argument"; return -1; }
    const int64_t L = h->read_len;
    int64_t at = 0;
    return run_stream(h, [&](uint8_t *dst, int64_t max_reads, int64_t *first) -> int64_t {
        const int64_t n = std::max<int64_t>(0, std::min(max_reads, nreads - at));
        if (n > 0) memcpy(dst, reads + at * L, (size_t)(n * L));
        *first = at; at += n;
        return n;
    }, first_read_id, nreads);
}

// process_seqfile + search_seqs + classify_reads over n_dev GPUs of this process (the library-owned form of the
// multi-GPU path; one process per GPU + RCCL is microbecensus_amd/distributed.py).  The sampler runs once; batches of
// MC_STREAM_BATCH accepted reads are dealt to the devices in the order they ask for them (reads are independent and keep their
// global ids), every device runs the streaming pipeline on a host thread of its own.  Results stay with the handles
// (mc_result_* per handle); the caller sums what it needs - the per-family accumulators are integers.
extern "C" int mc_search_files_multi(mc_handle *const *handles, int32_t n_dev, mc_reader *r, int64_t first_read_id)
{
    if (!handles || n_dev < 1 || !r) { g_err = "bad argument"; return -1; }
    for (int d = 0; d < n_dev; d++) {
        if (!handles[d] || !handles[d]->run_set) { g_err = "mc_set_run() must be called first on every handle"; return -1; }
        if (mc_reader_read_len(r) != handles[d]->read_len) { g_err = "the reader trims to another length than mc_set_run() was given"; return -1; }
    }
    if (mc_reader_start(r) != 0) { g_err = mc_reader_last_error(); return -1; }
    const int64_t cap_reads = mc_reader_nreads(r);                  // the reads the sampler may deliver at most (args['nreads']): per device, what to size for
    const int64_t expect = cap_reads > 0 && cap_reads < ((int64_t)1 << 40) ? (cap_reads + n_dev - 1) / n_dev : 0;
    std::mutex deal_mu;
    int64_t next = 0;
    bool ended = false;
    std::vector<int> rcs((size_t)n_dev, 0);
    std::vector<std::string> errs((size_t)n_dev);
    auto work = [&](int d) {
        std::string ferr;
        rcs[(size_t)d] = run_stream(handles[d], [&](uint8_t *dst, int64_t max_reads, int64_t *first) -> int64_t {
            int64_t at;
            { std::unique_lock<std::mutex> lk(deal_mu); if (ended) return 0; at = next; next += max_reads; }
            const int64_t n = mc_reader_fetch(r, at, max_reads, dst);
            if (n < 0) ferr = mc_reader_last_error();
            if (n < max_reads) { std::unique_lock<std::mutex> lk(deal_mu); ended = true; }
            *first = at;
            return n;
        }, first_read_id, expect);
        errs[(size_t)d] = !ferr.empty() ? ferr : std::string(rcs[(size_t)d] ? mc_last_error() : "");
    };
    std::vector<std::thread> th;
    for (int d = 1; d < n_dev; d++) th.emplace_back(work, d);
    work(0);
    for (auto &t : th) t.join();
    const int64_t sampled = mc_reader_join(r);
    for (int d = 0; d < n_dev; d++) if (rcs[(size_t)d] == -3) { g_err = errs[(size_t)d]; return -3; }
    if (sampled == -3) { g_err = mc_reader_last_error(); return -3; }
    for (int d = 0; d < n_dev; d++) if (rcs[(size_t)d]) { g_err = errs[(size_t)d]; return rcs[(size_t)d]; }
    if (sampled < 0) { g_err = mc_reader_last_error(); return (int)sampled; }
    return 0;
}

extern "C" int mc_search_files(mc_handle *h, mc_reader *r, int64_t first_read_id) { return mc_search_files_multi(&h, 1, r, first_read_id); }

extern "C" int mc_grid_classify(mc_handle *h, const double *aln_covs, int32_t n_cov, const int32_t *max_pids, int32_t n_pid, const double *min_scores, int32_t n_score,
                                int64_t *count_hits, int64_t *count_aln, double *count_cov)
{
    if (!h || !h->run_set) { g_err = "mc_set_run() must be called first"; return -1; }
    if (n_cov < 1 || n_cov > MC_GRID_MAXC || n_pid < 1 || n_pid > MC_GRID_MAXP || n_score < 1 || n_score > MC_GRID_MAXS) { g_err = "grid larger than 8 x 8 x 64"; return -1; }
    if (!aln_covs || !max_pids || !min_scores || !count_hits || !count_aln || !count_cov) { g_err = "null argument"; return -1; }
    HIPCK(hipSetDevice(h->device));
    const int nfam = h->nfam;
    McGridPars G; memset(&G, 0, sizeof G);
    G.read_len = h->read_len; G.n_cov = n_cov; G.n_pid = n_pid; G.n_score = n_score; G.nfam = nfam;
    for (int i = 0; i < n_cov; i++) G.cov[i] = aln_covs[i];
    for (int i = 0; i < n_pid; i++) G.pid[i] = max_pids[i];
    std::vector<int> order((size_t)n_score);
    for (int i = 0; i < n_score; i++) order[(size_t)i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return min_scores[a] < min_scores[b]; });
    for (int i = 0; i < n_score; i++) G.score[i] = min_scores[order[(size_t)i]];
    const size_t nbins = (size_t)n_cov * n_pid * (MC_GRID_MAXS + 1) * nfam;
    const size_t nout = (size_t)n_cov * n_pid * n_score * nfam;
    memset(count_hits, 0, nout * 8); memset(count_aln, 0, nout * 8); memset(count_cov, 0, nout * 8);
    const int64_t nrows = h->n_res_rows;
    if (nrows == 0) return 0;
    rows_wait(h);
    McRow *d_rows = nullptr; unsigned long long *d_bins = nullptr;
    HIPCK(hipMalloc((void **)&d_rows, (size_t)nrows * sizeof(McRow)));
    if (hipMalloc((void **)&d_bins, nbins * 24) != hipSuccess) { (void)hipFree(d_rows); g_err = "out of device memory"; return -1; }
    hipStream_t st = h->ctx[0].stream;
    HIPCK(hipMemcpyAsync(d_rows, h->res_rows, (size_t)nrows * sizeof(McRow), hipMemcpyHostToDevice, st));
    HIPCK(hipMemsetAsync(d_bins, 0, nbins * 24, st));
    k_grid_classify<<<dim3((unsigned)((nrows + 127) / 128)), dim3(128), 0, st>>>(G, dev_index(h), h->d_fam, d_rows, nrows, d_bins, d_bins + nbins, (double *)(d_bins + 2 * nbins));
    std::vector<unsigned long long> bins(nbins * 3);
    HIPCK(hipMemcpyAsync(bins.data(), d_bins, nbins * 24, hipMemcpyDeviceToHost, st));
    HIPCK(hipStreamSynchronize(st));
    (void)hipFree(d_rows); (void)hipFree(d_bins);
    const double *bcov = (const double *)(bins.data() + 2 * nbins);
    // bin nk = reads whose best row passes exactly the first nk (ascending) cut-offs: cut-off j (ascending) counts the bins nk > j
    for (int c = 0; c < n_cov * n_pid; c++)
        for (int f = 0; f < nfam; f++) {
            unsigned long long sh = 0, sa = 0; double sc = 0.0;
            for (int j = n_score - 1; j >= 0; j--) {
                const size_t o = ((size_t)c * (MC_GRID_MAXS + 1) + (size_t)(j + 1)) * (size_t)nfam + (size_t)f;
                sh += bins[o]; sa += bins[nbins + o]; sc += bcov[o];
                const size_t out = ((size_t)c * n_score + (size_t)order[(size_t)j]) * (size_t)nfam + (size_t)f;
                count_hits[out] = (int64_t)sh; count_aln[out] = (int64_t)sa; count_cov[out] = sc;
            }
        }
    return 0;
}

extern "C" int mc_set_keep_rows(mc_handle *h, int keep)
{
    if (!h) { g_err = "null handle"; return -1; }
    h->keep_rows = keep != 0;
    return 0;
}

extern "C" int mc_set_best_hits_only(mc_handle *h, int on)
{
    if (!h) { g_err = "null handle"; return -1; }
    h->best_only = on != 0;
    return 0;
}

extern "C" int64_t mc_result_rows(mc_handle *h, const mc_row **rows) { if (!h) return -1; rows_wait(h); *rows = h->res_rows; return h->n_res_rows; }
extern "C" int64_t mc_result_best_hits(mc_handle *h, const mc_best_hit **hits) { if (!h) return -1; *hits = h->best.data(); return (int64_t)h->best.size(); }
extern "C" int mc_result_stats(mc_handle *h, mc_stats *out) { if (!h) return -1; *out = h->stats; return 0; }

static int write_m8(mc_handle *h, const char *path, int append, const char *const *query_names, int64_t n_names, int64_t first_read_id)
{
    if (!h) { g_err = "null handle"; return -1; }
    rows_wait(h);
    FILE *f = fopen(path, append ? "a" : "w");
    if (!f) { g_err = std::string("cannot open ") + path; return -1; }
    setvbuf(f, nullptr, _IOFBF, 1 << 22);
    for (int64_t i = 0; i < h->n_res_rows; i++) {
        const mc_row &r = h->res_rows[i];
        if (query_names) {
            const int64_t k = (int64_t)r.query - first_read_id;
            if (k < 0 || k >= n_names) { fclose(f); g_err = "a row's query id lies outside the names given"; return -1; }
            fprintf(f, "%s", query_names[k]);
        } else fprintf(f, "%d", r.query);
        fprintf(f, "\t%s\t%g\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%g\t%g\n", h->H.names[r.subject].c_str(), r.ident, r.alnlen, r.mismatch, r.gapopen, r.qstart, r.qend,
                r.sstart, r.send, r.loge, r.bits);
    }
    fclose(f);
    return 0;
}
extern "C" int mc_write_m8(mc_handle *h, const char *path, int append) { return write_m8(h, path, append, nullptr, 0, 0); }
extern "C" int mc_write_m8_named(mc_handle *h, const char *path, int append, const char *const *query_names, int64_t n_names, int64_t first_read_id)
{
    if (!query_names) { g_err = "null names"; return -1; }
    return write_m8(h, path, append, query_names, n_names, first_read_id);
}
