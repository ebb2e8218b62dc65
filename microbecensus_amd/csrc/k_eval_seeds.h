// k_eval_seeds.h - stage A3: seed gate, growth and ungapped X-drop extension of the seed hits (ExtendSeq2Set@0x413b90,
// AlignFwd / AlignBwd) -> HSPs and gap tasks; the mark of the reads that can be classified (mc_set_best_hits_only).
#pragma once
#include "mc_hip_common.h"

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
#ifdef MC_EXP_TIMING
__device__ unsigned long long g_ev_turns[4];         // X-drop loops: lane-turns forward, wave-turns forward, lane-turns backward, wave-turns backward (a turn = 8 residues)
#define MC_EV_TURN(k) do { const unsigned long long m_ = __ballot(true); if ((int)__builtin_ctzll(m_) == (int)(threadIdx.x & 63)) { atomicAdd(&g_ev_turns[2 * (k)], (unsigned long long)__popcll(m_)); atomicAdd(&g_ev_turns[2 * (k) + 1], 1ull); } } while (0)
#else
#define MC_EV_TURN(k) do { } while (0)
#endif
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
#ifndef MC_EV_XDROP_STEPWISE
// One accumulator per walk carries what a step of AlignFwd / AlignBwd updates - the running score, the step's number and the identities
// so far: P = run * 65536 + (255 - i) * 256 + id (i <= MC_MAXAA steps, id <= i).  A step adds ONE table word to it (sub32[a][b] =
// score * 65536 - 256 + (a == b), 4 KB of LDS made at the kernel's start), and the best prefix is `max`: P' > Pbest <=> run > best
// (the low 16 bits are below 65536, and on equal scores the EARLIER step has the larger low part - the reference's `run > best` keeps the
// first one too).  best, its length and its identities are read off Pbest at the end.  The exit test best - run >= xdi is
// Pbest - P >= xdi * 65536 exactly (the low parts differ by (i - bl) * 256 - (id - bid) in [0, 65535]: i > bl whenever best > run).
// `run < -20` cannot fire first: best >= the seed's score >= MC_SEED_SCORE (the gate), so run < -20 has best - run > 31 >= xdi
// (static_assert below).  The end of the shorter sequence: the bytes of the query's word at and behind it are set to 0xFF - row 31 of the
// table, which no residue code uses (codes <= MC_INV = 20), holds - 100 * 65536: the step behind the last residue is an exit by the test
// that is there anyway, with best untouched.  Per step: three instructions for the table's address, one LDS read, add, max, subtract,
// compare, select - against fifteen and three exit tests (the step-by-step form, -DMC_EV_XDROP_STEPWISE: same results).
static_assert(MC_MAXAA <= 250 && MC_INV < 31 && MC_SEED_SCORE >= -11.0, "the packed accumulator of mc_ev_xdrop");
#define MC_EV_POISON (-100 * 65536 - 256)
#define MC_EV_SUB32(a, b) (*(const int32_t *)((const char *)sub32 + ((((a) << 7) | ((b) << 2)) & 0xFFCu)))      // (the word's byte address straight from the two bytes)
__device__ __forceinline__ void mc_ev_sub32(int32_t *sub32, const McHot &T)
{
    for (int x = (int)threadIdx.x; x < 1024; x += (int)blockDim.x) sub32[x] = (x >> 5) == 31 ? MC_EV_POISON : (int)T.sub[x] * 65536 - 256 + (int)((x >> 5) == (x & 31));
}
__device__ __forceinline__ int mc_ev_xdrop(const McHot &T, const int32_t *sub32, const uint8_t *q, int qlen, const uint8_t *d, int dlen, int sidx, int qp, int dp, int L, int score, int ident, McGapTask *gt)
{
    const int xdi = (int)floor(T.xdrop_ungapped) + 1;
    const int32_t TD = xdi * 65536, P0 = score * 65536 + 255 * 256;
    const int s0 = score;
    int qfwd = 0, qbwd = 0, fgain = 0, bgain = 0;
    { // forward
        const int n1 = qlen - qp - L, n2 = dlen - dp - L, lim = n1 < n2 ? n1 : n2;
        int32_t bestp = P0;
        if (lim > 0) {
            const uint8_t *p1 = q + qp + L, *p2 = d + dp + L;
            int32_t runp = P0;
            int i = 0;
            bool alive = true;
            do {
                MC_EV_TURN(0);
                uint64_t wa = mc_ld8(p1 + i);
                const uint64_t wb = mc_ld8(p2 + i);
                const int rem = lim - i;
                if (rem < 8) wa |= ~0ull << (8 * rem);
                int32_t e[8];
#pragma unroll
                for (int k = 0; k < 8; k++) e[k] = MC_EV_SUB32((uint32_t)(wa >> (8 * k)) & 0xFFu, (uint32_t)(wb >> (8 * k)) & 0xFFu);
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    runp += e[k];
                    const int32_t c = bestp > runp ? bestp : runp;
                    bestp = alive ? c : bestp;
                    alive = alive && (c - runp < TD);
                }
                i += 8;
            } while (alive && i < lim);
        }
        fgain = (bestp >> 16) - s0; qfwd = 255 - ((bestp >> 8) & 255); ident += bestp & 255;
    }
    { // backward, restarting from the seed score
        const int lim = qp < dp ? qp : dp;
        int32_t bestp = P0;
        if (lim > 0) {
            const uint8_t *p1 = q + qp - 8, *p2 = d + dp - 8;                      // residues a - 7 .. a of the step's first residue a = qp - 1 - i: step k uses byte 7 - k
            int32_t runp = P0;
            int i = 0;
            bool alive = true;
            do {
                MC_EV_TURN(1);
                uint64_t wa = mc_ld8(p1 - i);
                const uint64_t wb = mc_ld8(p2 - i);
                const int rem = lim - i;
                if (rem < 8) wa |= ~0ull >> (8 * rem);
                int32_t e[8];
#pragma unroll
                for (int k = 0; k < 8; k++) e[k] = MC_EV_SUB32((uint32_t)(wa >> (8 * (7 - k))) & 0xFFu, (uint32_t)(wb >> (8 * (7 - k))) & 0xFFu);
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    runp += e[k];
                    const int32_t c = bestp > runp ? bestp : runp;
                    bestp = alive ? c : bestp;
                    alive = alive && (c - runp < TD);
                }
                i += 8;
            } while (alive && i < lim);
        }
        bgain = (bestp >> 16) - s0; qbwd = 255 - ((bestp >> 8) & 255); ident += bestp & 255;
    }
    score = s0 + bgain + fgain;
    gt->sidx = (uint32_t)sidx; gt->qp = (int16_t)qp; gt->dp = (int16_t)dp; gt->L = (int16_t)L;
    gt->qfwd = (int16_t)qfwd; gt->qbwd = (int16_t)qbwd; gt->score = (int16_t)score; gt->nmatch = (int16_t)ident;
    return (!(T.gap_trigger > (double)score)) ? 2 : 1;
}
#else
__device__ __forceinline__ void mc_ev_sub32(int32_t *, const McHot &) { }
__device__ __forceinline__ int mc_ev_xdrop(const McHot &T, const int32_t *, const uint8_t *q, int qlen, const uint8_t *d, int dlen, int sidx, int qp, int dp, int L, int score, int ident, McGapTask *gt)
{
    // The reference's exit test is `(double)run < (double)best - xdrop` on two integers and a constant that is no integer (8.94 for BLOSUM62's
    // ungapped lambda; mc_fill_tables refuses one that is nearer than 1e-6 to an integer): best - run > xdrop <=> best - run >= floor(xdrop) + 1,
    // exactly.  (MC_EV_F64_XDROP: the reference's own form - two conversions and an f64 subtraction in the dependent chain of every step.)
#ifdef MC_EV_F64_XDROP
    const double xd = T.xdrop_ungapped;
#define MC_EV_DROP(run, best) ((double)(run) < (double)(best) - xd)
#else
    const int xdi = (int)floor(T.xdrop_ungapped) + 1;
#define MC_EV_DROP(run, best) ((best) - (run) >= xdi)
#endif
    int s0 = score, qfwd = 0, qbwd = 0, fgain = 0, bgain = 0;
    { // forward
        const int n1 = qlen - qp - L, n2 = dlen - dp - L;
        int bl = 0, bi = 0;
        if (n1 != 0 && n2 != 0 && !(s0 < -20)) {
            const uint8_t *p1 = q + qp + L, *p2 = d + dp + L;
            int run = s0, best = s0, id = 0, i = 0;
            bool stop = false;
            do {
                MC_EV_TURN(0);
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
                        stop = !(n2 > i) || n1 <= i || run < -20 || MC_EV_DROP(run, best);
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
                MC_EV_TURN(1);
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
                        stop = b < 0 || a < 0 || run < -20 || MC_EV_DROP(run, best);
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
#endif

#ifdef MC_EXP_TIMING
__device__ unsigned long long g_ev_acc[8];           // wave time per phase, summed over the waves (lane 0): 0 loop, next records asked for 1 survivors queued 2 X-drop extension (with the read of the queue) 3 HSP, marks 4 records written 5 wait for the residues + seed score 6 redundancy test, growth, gate
#define MC_EV_TICK(prev) do { const unsigned long long now_ = __builtin_readcyclecounter(); ev_acc_[prev] += now_ - ev_last_; ev_last_ = now_; } while (0)
#else
#define MC_EV_TICK(prev) do { } while (0)
#endif

#define MC_EV_BS 256         // threads per workgroup (the waves are on their own: the size only sets how the LDS is handed out)
#define MC_EV_BPC 5          // workgroups per CU: 20 waves, 5 per SIMD - 88 registers, nothing spilled (measured per 1 M reads of 150 / 300 bp:
                             // 7 waves per SIMD and 72 registers with 52 bytes of scratch 3.50 / 7.77 ms, 6 with 80 and 12 bytes 2.82 / 6.41, 5 with 88 2.60 / 6.01, 4: 2.79 / 6.53)
                             // (again after the postings moved here - <true>, 85 registers: 6 waves per SIMD 2.89 / 6.29, 5: 2.57 / 5.67, 4: 2.66 / 6.06)
#define MC_EV_QCAP 128       // survivors of the gate a wave holds (32 bytes each: 4 KB of LDS per wave)
#define MC_EV_BLK 256u       // slots of the HSP / gap-task pools a wave reserves at a time (one global atomic per block)
#ifndef MC_EV_GROUP
#define MC_EV_GROUP 16u      // chunks of 64 hits a wave takes at a time (one global atomic per group)
#endif
static_assert(MC_EV_GROUP >= 2, "the next group is asked for while the last but one chunk of a group is evaluated (cleft == 1): with one chunk per group no request is ever made and every wave would walk the same hits");
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
// RANGES (round 5, the product's seed kernel k_enumerate_q; the name is history: ranges of hits per record were built first and cost this
// kernel more than they saved the other): a record holds the INDEX of its hit's posting instead of the posting, its place in the residue
// array and the rest of the subject - the seed kernel is bound by the scattered lines its CUs fetch and a hit's posting and
// offsets were 150 of the 810 lines a read cost it; this kernel was thought not to wait for memory (5.7; it does: MC_POST_WORDS below) and fetches the three in one 8-byte load.
// the postings with what the evaluation needs beside them (MC_POST8), made once per handle
// MC_POST_WORDS = 4 (round 6): the 24 residues of the subject around the posting - dpos - 8 .. dpos + 15, what the gate reads of it - travel
// WITH it: 32 bytes per posting (115 MB for the marker database instead of 29 + the 14 MB residue array asked at a scattered place), one
// aligned item = ONE line per hit instead of 2.4, and no second trip (posting -> place -> residues).  The kernel takes the time the memory
// system needs for its scattered lines (instructions, occupancy and loads ahead of time all left it where it was), so lines are
// what pays.  The X-drop walks of the three hits in ten that pass the gate still read the residue array.  -DMC_POST_WORDS=1: the 8-byte form.
#ifndef MC_POST_WORDS
#define MC_POST_WORDS 4
#endif
__global__ void __launch_bounds__(256) k_post8(const uint32_t *__restrict__ post, const uint32_t *__restrict__ off, const uint8_t *__restrict__ res, uint32_t n, unsigned long long *post8)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t pst = post[i], s = pst >> 11, abs = off[s] + (pst & 0x7ffu);
    post8[(size_t)i * MC_POST_WORDS] = MC_POST8(pst, abs, off[s + 1] - abs);
#if MC_POST_WORDS == 4
    post8[(size_t)i * 4 + 1] = mc_ld8(res + abs - 8); post8[(size_t)i * 4 + 2] = mc_ld8(res + abs); post8[(size_t)i * 4 + 3] = mc_ld8(res + abs + 8);   // (the residue array has room on both sides)
#endif
}
template <bool RANGES>
__global__ void __attribute__((amdgpu_waves_per_eu(5, 5))) __launch_bounds__(MC_EV_BS) k_eval_seeds(const McTables *__restrict__ T, McIndex X, const uint8_t *__restrict__ frames, int FP, int L,
                                                    const McSeedTask *__restrict__ tasks, const uint32_t *__restrict__ ntasks_p, uint32_t cap_tasks, McHsp *hsps, uint32_t cap_hsps,
                                                    McGapTask *gaps, uint32_t cap_gaps, uint32_t *counters, const McClassPars *__restrict__ P, const int32_t *__restrict__ fam, uint8_t *cand, uint64_t *hkeys, uint8_t *low, uint64_t *hplace)
{
    const uint32_t ntasks = *ntasks_p <= cap_tasks ? *ntasks_p : 0u;   // (device-side count of the seed kernel; after an overflow the host discards the batch)
    __shared__ McHot hot;
#ifndef MC_EV_XDROP_STEPWISE
    __shared__ int32_t sub32[1024];
#else
    int32_t *const sub32 = nullptr;
#endif
    const int lane = mc_lane(), wv = threadIdx.x >> 6;
    uint4 *Q = (uint4 *)(mc_smem + (size_t)wv * MC_EV_QCAP * 32);    // entry e: words 2 e, 2 e + 1
    mc_load_hot(&hot, T);
    __syncthreads();
    mc_ev_sub32(sub32, hot);
    __syncthreads();
    const double hot_loge_thr = T->loge_thr;
    uint32_t qn = 0, hb_base = 0, hb_used = MC_EV_BLK, gb_base = 0, gb_used = MC_EV_BLK;
    bool ok = true;
    const unsigned long long lt = (1ull << lane) - 1;
    // Which records a wave takes: groups of MC_EV_GROUP chunks of 64 consecutive records, the first group by the wave's number, every further
    // one from a counter (asked for one chunk ahead of need) - dealt out in turn, the launch lasted as long as the wave whose hits
    // took longest (the same finding as in k_enumerate_t0: reads of marker genes cost many times the average).
    const uint32_t nchunks = (ntasks + 63u) / 64u, nwaves = gridDim.x * (MC_EV_BS / 64);
    uint32_t cur = (blockIdx.x * (MC_EV_BS / 64) + (uint32_t)wv) * MC_EV_GROUP, cleft = MC_EV_GROUP - 1, pend = 0;
    // The chain of dependent reads of a hit was: its record -> the subject's offsets -> the residue in front of the seed -> the
    // seed's residues, four trips to the L2 before the gate.  Now: the record of the NEXT chunk is fetched while this one is
    // evaluated, and the residues in front of the seed and the seed's own ten are read together.  (!RANGES: the record of a single hit
    // carries the hit's position in the residue array - MC_TASK_W3 - and what is left of the subject behind it - MC_TASK_READ.)
#ifdef MC_EXP_TIMING
    unsigned long long ev_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ev_last_ = __builtin_readcyclecounter();
#endif
    // ---- phase 1 for one hit per lane: the gate; the survivors into the wave's queue
    auto gate = [&](bool active, uint32_t rd, uint32_t chrono, uint32_t posting, uint32_t abs, int rem, int seedlen, int nkey, uint64_t r0, uint64_t r1, uint64_t r2) {
        bool surv = false;
        uint4 e0 = {0, 0, 0, 0}, e1 = {0, 0, 0, 0};
        if (active) {
            const int frame = (int)(chrono >> 25), pos = (int)((chrono >> 17) & 0xff);
            const int qlen = (L - frame % 3) / 3;
            const int dpos = (int)(posting & 0x7ffu), sidx = (int)(posting >> 11);
            const uint32_t o0 = abs - (uint32_t)dpos;
            const uint8_t *q = frames + ((int64_t)rd * 6 + frame) * FP, *d = X.res + o0;
            // residues pos - 8 .. pos + 15 of the frame and dpos - 8 .. dpos + 15 of the subject: six loads, one trip (rows and residue array have room on both sides)
            const uint64_t q0 = mc_ld8(q + pos - 8), q1 = mc_ld8(q + pos), q2 = mc_ld8(q + pos + 8);
            uint64_t d0 = r0, d1 = r1, d2 = r2;                   // (with the posting's record: MC_POST_WORDS = 4)
            if (!(RANGES && MC_POST_WORDS == 4)) { d0 = mc_ld8(d + dpos - 8); d1 = mc_ld8(d + dpos); d2 = mc_ld8(d + dpos + 8); }
            const int qm1 = (int)(q0 >> 56), dm1 = (int)(d0 >> 56);
            const int dlen = dpos + rem;
            int score = 0, ident = 0;
#pragma unroll
            for (int k = 0; k < 10; k++)
                if (k < seedlen) {
                    const int a = (int)((k < 8 ? q1 >> (8 * k) : q2 >> (8 * (k - 8))) & 0xFFu), b = (int)((k < 8 ? d1 >> (8 * k) : d2 >> (8 * (k - 8))) & 0xFFu);
                    score += MC_SUB(hot, a, b); ident += (a == b);
                }
            MC_EV_TICK(5);                                            // (the six residue loads arrive here: their wait, and the seed's score)
            const bool go = !(dpos + seedlen > dlen) && !(pos != 0 && dpos != 0 && hot.grp[qm1] == hot.grp[dm1] && nkey != 4);
            int qp = 0, dp = 0, Lg = 0;
            if (go) surv = mc_ev_gate(hot, q, qlen, pos, d, dlen, dpos, seedlen, score, ident, qp, dp, Lg, q0, q2, d0, d2);
            MC_EV_TICK(6);                                            // (redundancy test, growth, the gate's thresholds)
            e0.x = rd; e0.y = chrono; e0.z = o0; e0.w = (uint32_t)sidx;
            e1.x = (uint32_t)qp | ((uint32_t)dp << 16); e1.y = (uint32_t)Lg | ((uint32_t)(uint16_t)(int16_t)score << 16); e1.z = (uint32_t)ident | ((uint32_t)dlen << 16);
        }
        const unsigned long long ms = __ballot(surv);
        if (surv) { const uint32_t at = qn + (uint32_t)__popcll(ms & lt); Q[2 * at] = e0; Q[2 * at + 1] = e1; }
        qn += (uint32_t)__popcll(ms);
        mc_wave_sync();
    };
    // ---- phase 2: the extension, up to 64 survivors of the queue
    auto extend = [&]() {
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
            rc = mc_ev_xdrop(hot, sub32, q, qlen, d, (int)(e1.z >> 16), sidx, (int)(e1.x & 0xFFFFu), (int)(e1.x >> 16), (int)(e1.y & 0xFFFFu), (int)(int16_t)(e1.y >> 16), (int)(e1.z & 0xFFFFu), &g);
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
            else if (keep) { mc_store_stream(&hsps[slot], h); __builtin_nontemporal_store(MC_HSP_KEY(h), &hkeys[slot]); __builtin_nontemporal_store(MC_HSP_PLACE(h), &hplace[slot]); }
        }
        if (mg && ok) {
            const uint32_t slot = mc_ev_slots((uint32_t)__popcll(mg), (uint32_t)__popcll(mg & lt), cap_gaps, &counters[C_GAPS], gb_base, gb_used, &ok, lane);
            if (!ok) { if (lane == 0) counters[C_OVERFLOW] = 3; }
            else if (rc == 2) mc_store_stream(&gaps[slot], g);
        }
        MC_EV_TICK(4);
    };
    McSeedTask tn;
    tn.read = MC_TASK_NONE; tn.chrono = 0; tn.posting = 0; tn.seedlen_nkey = 0;
    if ((uint64_t)cur * 64 + (uint32_t)lane < ntasks) tn = mc_load_stream(&tasks[(uint64_t)cur * 64 + (uint32_t)lane]);
    // the chunk after the current one: its number, and the request for the group after this one (one chunk ahead of need)
    auto advance = [&]() -> uint32_t {
        uint32_t nxt = cur + 1;
        if (cleft == 1 && lane == 0) pend = atomicAdd(&counters[C_EVCHUNK], 1u);
        if (cleft > 0) cleft--;
        else { nxt = (nwaves + (uint32_t)__builtin_amdgcn_readfirstlane((int)pend)) * MC_EV_GROUP; cleft = MC_EV_GROUP - 1; }
        return nxt;
    };
    if constexpr (!RANGES) {
        for (;;) {
            const bool last = cur >= nchunks;
            if (!last) {
                MC_EV_TICK(0);
                const uint32_t tid = cur * 64u + (uint32_t)lane;
                const McSeedTask t = tn;
                {
                    const uint32_t nxt = advance();
                    const uint64_t nx = (uint64_t)nxt * 64 + (uint32_t)lane;
                    tn.read = MC_TASK_NONE;
                    if (nx < ntasks) tn = mc_load_stream(&tasks[nx]);
                    cur = nxt;
                }
                const uint32_t w3 = t.seedlen_nkey;
                gate(tid < ntasks && t.read != MC_TASK_NONE, MC_TASK_READ_OF(t.read), t.chrono, t.posting, w3 & 0xFFFFFFu, (int)MC_TASK_REM_OF(t.read), (int)((w3 >> 24) & 15u), (int)(w3 >> 28), 0, 0, 0);   // (MC_TASK_NONE: padding of a partly used block of the pool)
                MC_EV_TICK(1);
            }
            while (qn >= 64 || (last && qn > 0)) extend();
            if (last) break;
        }
    } else {
        // The record holds the INDEX of the hit's posting: posting, position in the residue array and rest of the subject come in one 8-byte
        // load (MC_POST8) - asked for one chunk ahead (the records two chunks ahead), so that a chunk still makes one trip: the residues.
        McSeedTask tn2;
        tn2.read = MC_TASK_NONE; tn2.chrono = 0; tn2.posting = 0; tn2.seedlen_nkey = 0;
        uint32_t tidn = cur * 64u + (uint32_t)lane, tidn2 = 0;       // the hit numbers of tn / tn2 (past the pool: no hit)
        uint32_t cur2 = cur;
        bool have_n = cur < nchunks, have_n2 = false;
        unsigned long long p8n = 0, r0n = 0, r1n = 0, r2n = 0;
        auto fetch = [&](uint32_t idx) {
#if MC_POST_WORDS == 4
            const mc_u32x4 *rp = (const mc_u32x4 *)(X.post8 + (size_t)idx * 4);
            const mc_u32x4 lo = rp[0], hi = rp[1];                // (not "nontemporal": the hits of a probe have neighbouring records - past the caches 2.12 -> 2.56 ms per 1 M reads)
            p8n = (unsigned long long)lo.x | ((unsigned long long)lo.y << 32); r0n = (unsigned long long)lo.z | ((unsigned long long)lo.w << 32);
            r1n = (unsigned long long)hi.x | ((unsigned long long)hi.y << 32); r2n = (unsigned long long)hi.z | ((unsigned long long)hi.w << 32);
#else
            p8n = X.post8[idx];
#endif
        };
        if (have_n) {
            cur2 = advance(); have_n2 = cur2 < nchunks; tidn2 = cur2 * 64u + (uint32_t)lane;
            if (have_n2 && tidn2 < ntasks) tn2 = mc_load_stream(&tasks[tidn2]);
            fetch((tidn < ntasks && tn.read != MC_TASK_NONE) ? tn.posting : 0u);
        }
        for (;;) {
            const bool last = !have_n;
            if (!last) {
                MC_EV_TICK(0);
                const McSeedTask t = tn;
                const unsigned long long p8 = p8n, r0 = r0n, r1 = r1n, r2 = r2n;
                const bool have = tidn < ntasks && t.read != MC_TASK_NONE;   // (MC_TASK_NONE: padding of a partly used block of the pool)
                // the chunks behind: tn2 (arrived) becomes tn and its postings are asked for; the records of the chunk after it are asked for
                tn = tn2; tidn = tidn2; have_n = have_n2;
                if (have_n) {
                    fetch((tidn < ntasks && tn.read != MC_TASK_NONE) ? tn.posting : 0u);
                    cur = cur2; cur2 = advance(); have_n2 = cur2 < nchunks; tidn2 = cur2 * 64u + (uint32_t)lane;
                    tn2.read = MC_TASK_NONE;
                    if (have_n2 && tidn2 < ntasks) tn2 = mc_load_stream(&tasks[tidn2]);
                }
                const uint32_t w3 = t.seedlen_nkey;
                gate(have, MC_TASK_READ_OF(t.read), t.chrono, (uint32_t)p8 & 0x3FFFFFFu, (uint32_t)(p8 >> 26) & 0xFFFFFFu, (int)(p8 >> 50), (int)((w3 >> 24) & 15u), (int)(w3 >> 28), r0, r1, r2);
                MC_EV_TICK(1);
            }
            while (qn >= 64 || (last && qn > 0)) extend();
            if (last) break;
        }
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
    if (lane == 0) for (int k = 0; k < 7; k++) atomicAdd(&g_ev_acc[k], ev_acc_[k]);
#endif
}
