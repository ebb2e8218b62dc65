// mc_core.h - per-thread algorithms of the MI355X translated-search path.
//
// Every function here is what ONE GPU thread (or one wave, where noted) executes; the HIP kernels in
// mc_kernels.hip are thin launch wrappers around them.  The functions are MC_HD so that the very same
// source can also be compiled by g++ into the test-only emulation harness (tests/emul/) that checks the
// logic against the oracle on machines without a GPU.  The product never runs that harness.
//
// What is computed follows RAPsearch2 v2.15 as the reference drives it
// (/root/reference/microbe_census/microbe_census.py:369-389: rapsearch -z T -e 1 -t n -p f -b 0); the
// addresses quoted are inside /root/reference/microbe_census/bin/rapsearch_Linux_2.15 and are the same
// ones oracle/rapsearch_port.c documents.  This file is an independent, GPU-shaped formulation:
// dense 5-bit residue codes, a 32x32 int8 score tile, table-driven statistics, per-candidate task
// records and stateless (trace-free) gapped extension.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MC_HD __host__ __device__ __forceinline__
#define MC_HDN __host__ __device__
#else
#define MC_HD inline
#define MC_HDN inline
struct uint4 { uint32_t x, y, z, w; };
inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { uint4 v; v.x = x; v.y = y; v.z = z; v.w = w; return v; }
#endif

// ---------------------------------------------------------------------------------------------
// constants
// ---------------------------------------------------------------------------------------------
#define MC_NBUCKET 1000000
#define MC_INV 20          // dense code of every non-amino-acid residue (stop, unknown codon, SEG mask)
#define MC_INVGRP 10       // reduced-alphabet group of MC_INV
#define MC_MAXAA 170       // longest frame supported (read length <= 510)
#define MC_SMAX 4096       // raw scores are tabulated up to here
#define MC_MAX_M8 500      // rapsearch -v 500
#define MC_LNFAC_N 400

#define MC_GAP_OPEN 11
#define MC_GAP_EXT 1
#define MC_SEG_FXBITS 24
#define MC_SEED_SCORE 11.0
#define MC_SEED_IDENT 4

// chronological key of a seed hit inside one read: frame(3) | pos(8) | phase(6) | posting(11)
#define MC_CHRONO(frame, pos, phase, idx) ((((uint32_t)(frame)) << 25) | (((uint32_t)(pos)) << 17) | (((uint32_t)(phase)) << 11) | ((uint32_t)(idx)))

struct McTables {
    int8_t sub[32 * 32];       // BLOSUM62 on dense codes, -5 wherever MC_INV is involved (this+0x31c)
    uint8_t grp[32];           // dense code -> murphy10 group (this+0x219)
    uint8_t codon[64];         // 16*i(b0)+4*i(b1)+i(b2), T,C,A,G = 0..3 -> dense code (aa@0x688c00)
    double entray[2][16];      // Seg::entropy_init@0x439090 for W = 12 / 8
    double lterm[13][16];      // log((1/total)*c)*c, Seg::entropy_cal@0x438f70 general branch
    double lnfac[MC_LNFAC_N];  // lnfac@0x688c40
    // fixed-point form of the window-entropy tests (mc_seg_mask_fx): seg_f[c] = c*log2(c), seg_tlo/thi[t] = t*(log2(t) - 2.2 / 2.5),
    // all scaled by 2^MC_SEG_FXBITS; seg_dout[c] = seg_f[c-1] - seg_f[c], seg_din[c] = seg_f[c+1] - seg_f[c]
    int32_t seg_dout[16], seg_din[16], seg_tlo[16], seg_thi[16];   // contiguous: read as one array of 64
    double xdrop_ungapped, xdrop_gapped, gap_trigger;   // this+0x40388, +0x40398, +0x40378
    // per run (query length m = read_len/3): BlastStat state after blastComputeLengthAdjustmentComp
    double ell, mprime, nprime, logK;
    double loge_thr;           // -e 1
    double loge_r[MC_SMAX];    // CalRes 0x4077ec-0x407b8f: rounded log10(E) per raw score
    double bits_r[MC_SMAX];    // CalRes 0x40783f-0x407850 / 0x407be8: rounded bit score per raw score
    uint32_t freq_thr;         // .info median bucket size (Db+0x38)
    double letter_p[10];       // Db+0x28
};

// the members of McTables that the extension kernels read per residue, small enough for LDS (1.1 KB)
struct McHot {
    int8_t sub[32 * 32];
    uint8_t grp[32];
    double xdrop_ungapped, xdrop_gapped, gap_trigger;
};

struct McIndex {
    const uint8_t *res;        // dense residue codes of all marker sequences
    const uint32_t *off;       // nseq+1
    const uint32_t *bstart;    // MC_NBUCKET+1
    const uint32_t *post;      // seqIdx<<11 | pos, in prerapsearch bucket order
    const unsigned long long *post8;   // (device) the same postings with what the evaluation of a hit needs beside them: posting 26 bits | position in the residue array 24 << 26 | what is left of the subject from there 11 << 50 (MC_POST8)
    const uint16_t *keys;      // 4 reduced residues after the 6-mer, 0xF past the sequence end
    const struct McBucketRec *rec;   // per bucket: start + first-residue group boundaries (NULL when the index cannot use them)
    const uint32_t *wild;      // MC_WILD_LINES x 8 words: wildcard filter (mc_wild_*)
    const uint32_t *pair;      // MC_PAIR_BLOCKS x 4 words: which residues at a wildcard offset complete an index 10-mer (mc_pair_*)
    const unsigned long long *rt; uint32_t rt_mask;   // range table of the long groups (mc_rt_*): rt_mask + 1 slots
    const uint32_t *filt;      // MC_FILT9_WORDS words: Bloom filter over the (bucket, 3-residue key) pairs of the index (exact 9-mer probes)
    int32_t nseq;
};

// one seed hit waiting for evaluation (ExtendSeq2Set body for one posting)
struct McSeedTask {
    uint32_t read;             // read index inside the batch
    uint32_t chrono;           // MC_CHRONO
    uint32_t posting;
    uint32_t seedlen_nkey;     // seedlen | nkey<<8 (emulation) / MC_TASK_W3 (kernels)
};
// What the kernels keep in the fourth word: the hit's position in the residue array (off[sidx] + dpos, 24 bits: the marker
// database has 5.5 M residues; mc_set_db refuses one past 16 M), so that the evaluation reads the subject's residues without
// first fetching the subject's offset; seed length and number of key residues above it.
#define MC_TASK_W3(abs, seedlen, nkey) ((uint32_t)(abs) | ((uint32_t)(seedlen) << 24) | ((uint32_t)(nkey) << 28))
// ... and, in the kernels' records, the upper 11 bits of `read` hold what is left of the subject from the hit's position on (its length
// minus the position: 1 .. 2047) - the seed kernel has the subject's two offsets in one load, and the evaluation kernel then reads
// nothing of the index but residues (round 5: one scattered line less per hit; it is bound by the lines its CU's L1 has to fetch)
#define MC_TASK_READ(r, rem) ((uint32_t)(r) | ((uint32_t)(rem) << 21))
#define MC_TASK_READ_OF(w) ((w) & 0x1FFFFFu)
#define MC_TASK_REM_OF(w) ((w) >> 21)
static_assert(MC_TASK_READ_OF(0xFFFFFFFFu) == 0x1FFFFFu, "the padding record (read = all ones) is no read of a batch: a batch holds at most 2,097,151 reads");
#define MC_TASK_ABS_LIMIT (1u << 24)
#define MC_POST8(pst, abs, rem) ((unsigned long long)(pst) | ((unsigned long long)(abs) << 26) | ((unsigned long long)(rem) << 50))   // (posting < 2^26: at most 32,767 sequences of at most 2,047 residues)

// seed hit whose ungapped score reached the gapped trigger (AlignSeqs 0x4134c8)
struct McGapTask {
    uint32_t read, chrono, sidx;
    int16_t qp, dp, L, qfwd, qbwd;   // grown seed start (frame / subject), length, ungapped extents
    int16_t score, nmatch;
};

// one HSP (STResult as CalRes@0x4077a0 fills it)
struct McHsp {
    uint32_t read, chrono;
    int32_t sidx;
    int16_t score, frame;
    int16_t alnlen, mism, gaps, nmatch;
    int16_t qaas, qaae, ds, de;
    int16_t qnts, qnte;
    double loge;               // may be replaced by sum statistics during finishing
};

// m8 row (PrintRes@0x409310, second loop)
struct McRow {
    int32_t query, subject;
    double ident;
    int32_t alnlen, mismatch, gapopen, qstart, qend, sstart, send;
    double loge, bits;
    int32_t score, frame;
};

// ---------------------------------------------------------------------------------------------
// 6-frame translation (BuildQHash 0x40d079-0x40d433) - one thread per (read, frame)
// ---------------------------------------------------------------------------------------------
MC_HD int mc_nt_idx(uint8_t c) { return c == 'T' ? 0 : c == 'C' ? 1 : c == 'A' ? 2 : c == 'G' ? 3 : -1; }
MC_HD int mc_nt_idx_rc(uint8_t c)
{ // index of the complement; the std::map of Process knows ACGTU (+lower case, which never translates)
    return c == 'A' ? 0 : c == 'G' ? 1 : (c == 'T' || c == 'U') ? 2 : c == 'C' ? 3 : -1;
}

// The same two maps without compare chains: (c >> 1) & 3 separates the upper-case bases (A 0, C 1, T and U 2, G 3); a 32-bit set
// says which of 'A'..'`' are bases at all, a packed permutation turns the separated code into the index.
#define MC_NT_FWD_SET 0x00080045u    // A C G T
#define MC_NT_FWD_PERM 0xC6u         // A -> 2, C -> 1, T -> 0, G -> 3
#define MC_NT_RC_SET 0x00180045u     // A C G T U
#define MC_NT_RC_PERM 0x6Cu          // A -> 0, C -> 3, T/U -> 2, G -> 1
MC_HD int mc_nt_code(uint32_t c, uint32_t set, uint32_t perm)
{
    const uint32_t k = c - 0x41u;
    return (k < 32u && ((set >> k) & 1u)) ? (int)((perm >> (((c >> 1) & 3u) * 2)) & 3u) : -1;
}

MC_HD int mc_translate_frame(const McTables &T, const uint8_t *read, int len, int frame, uint8_t *prot)
{
    int o = frame % 3, n = (len - o) / 3, i;
    if (n < 0) n = 0;
    if (frame < 3) {
        for (i = 0; i < n; i++) {
            int a = mc_nt_idx(read[o + 3 * i]), b = mc_nt_idx(read[o + 3 * i + 1]), c = mc_nt_idx(read[o + 3 * i + 2]);
            prot[i] = (a < 0 || b < 0 || c < 0) ? MC_INV : T.codon[16 * a + 4 * b + c];
        }
    } else {
        for (i = 0; i < n; i++) {
            int p = len - 1 - (o + 3 * i);
            int a = mc_nt_idx_rc(read[p]), b = mc_nt_idx_rc(read[p - 1]), c = mc_nt_idx_rc(read[p - 2]);
            prot[i] = (a < 0 || b < 0 || c < 0) ? MC_INV : T.codon[16 * a + 4 * b + c];
        }
    }
    return n;
}

// ---------------------------------------------------------------------------------------------
// SEG (Seg::*@0x438b50-0x43b140; window W, locut 2.2, hicut 2.5, maxtrim 100, downset 0, upset 1)
// Residues are dense codes; MC_INV does not count towards the composition.  Positions of masked
// residues are returned as a bit set.
// ---------------------------------------------------------------------------------------------
MC_HD void mc_seg_state(const uint8_t *comp, uint8_t *sv)
{ // composition counts sorted descending, 0 terminated (Seg::stateon@0x4399b0)
    int n = 0;
    for (int i = 0; i < 20; i++) {
        int v = comp[i];
        if (v > 0) {
            int j = n;
            while (j > 0 && sv[j - 1] < v) { sv[j] = sv[j - 1]; j--; }
            sv[j] = (uint8_t)v; n++;
        }
    }
    sv[n] = 0;
}
MC_HD double mc_seg_entropy(const McTables &T, int W, const uint8_t *sv)
{ // Seg::entropy_cal@0x438f70
    int total = 0, i;
    double ent = 0.0;
    for (i = 0; sv[i]; i++) total += sv[i];
    if (total == W) { const double *e = T.entray[W == 12 ? 0 : 1]; for (i = 0; sv[i]; i++) ent = ent + e[sv[i]]; return ent; }
    if (total == 0) return 0.0;
    for (i = 0; sv[i]; i++) ent = ent + T.lterm[total][sv[i]];
    double inv = 1.0 / (double)total;
    double r = inv * (-ent);
    return r / 0.6931471805599453;
}
MC_HD void mc_seg_comp(const uint8_t *s, int n, uint8_t *comp)
{
    for (int i = 0; i < 20; i++) comp[i] = 0;
    for (int i = 0; i < n; i++) if (s[i] < 20) comp[s[i]]++;
}
// The same counts gathered in registers (20 classes x 8 bits in three words) and written once: with the counts in memory every
// residue is a read of the residue, a read of its count and a write of it, each waiting for the one before - a window of 60
// residues costs more than the eight probabilities of a trimming item behind it.  comp must be 4-byte aligned.
MC_HD void mc_seg_comp_rg(const uint8_t *s, int n, uint8_t *comp)
{
    uint64_t c0 = 0, c1 = 0;
    uint32_t c2 = 0;
    int i = 0;
    for (; i + 4 <= n; i += 4) {
        const int r0 = s[i], r1 = s[i + 1], r2 = s[i + 2], r3 = s[i + 3];
#define MC_COMP_ADD(r) do { const uint64_t inc = 1ull << (((r) & 7) * 8); if ((r) < 8) c0 += inc; else if ((r) < 16) c1 += inc; else if ((r) < 20) c2 += (uint32_t)inc; } while (0)
        MC_COMP_ADD(r0); MC_COMP_ADD(r1); MC_COMP_ADD(r2); MC_COMP_ADD(r3);
    }
    for (; i < n; i++) { const int r = s[i]; MC_COMP_ADD(r); }
#undef MC_COMP_ADD
    uint32_t *w = (uint32_t *)comp;
    w[0] = (uint32_t)c0; w[1] = (uint32_t)(c0 >> 32); w[2] = (uint32_t)c1; w[3] = (uint32_t)(c1 >> 32); w[4] = c2;
}
MC_HD double mc_seg_getprob(const double *lnfac, const uint8_t *sv, int total)
{ // Seg::getprob@0x4393d0 = lnperm + lnass - total*ln 20
    double ans1 = lnfac[20];
    if (sv[0] != 0) {
        int tot = 20, cls = 1, svim1 = sv[0], svi = 0, i;
        for (i = 0;; svim1 = svi) {
            if (++i == 20) { ans1 = ans1 - lnfac[cls]; break; }
            else if ((svi = sv[i]) == svim1) cls++;
            else {
                tot -= cls;
                ans1 = ans1 - lnfac[cls];
                if (svi == 0) { ans1 = ans1 - lnfac[tot]; break; }
                cls = 1;
            }
        }
    }
    double ans2 = lnfac[total];
    for (int i = 0; sv[i] != 0; i++) ans2 = ans2 - lnfac[sv[i]];
    double t = (double)total * 2.995732273553991;
    return (ans2 + ans1) - t;
}
MC_HD void mc_seg_trim(const McTables &T, const uint8_t *s, int n, int *leftend, int *rightend)
{ // Seg::trim@0x439e20
    int lend = 0, rend = n - 1, minlen = 1;
    double minprob = 1.0;
    uint8_t comp[20], sv[21];
    if (n - 100 > minlen) minlen = n - 100;
    for (int len = n; len > minlen; len--) {
        mc_seg_comp(s, len, comp);
        for (int i = 0;; i++) {
            mc_seg_state(comp, sv);
            double prob = mc_seg_getprob(T.lnfac, sv, len);
            if (prob < minprob) { minprob = prob; lend = i; rend = len + i - 1; }
            if (i + 1 + len > n) break;
            if (s[i] < 20) comp[s[i]]--;
            if (s[i + len] < 20) comp[s[i + len]]++;
        }
    }
    *leftend = *leftend + lend;
    *rightend = *rightend - (n - rend - 1);
}
// Seg::segseq@0x43a9e0 with the recursion turned into a work list (the result is a set of positions,
// so the order in which sub-ranges are handled does not matter).  H must hold n doubles.
MC_HDN void mc_seg_mask(const McTables &T, const uint8_t *prot, int n, uint8_t *maskbits /* (MC_MAXAA+7)/8 */, double *H)
{
    int W = (n <= 11) ? 8 : 12;
    int16_t stk_s[16], stk_n[16];
    int sp = 0;
    for (int i = 0; i < (MC_MAXAA + 7) / 8; i++) maskbits[i] = 0;
    stk_s[0] = 0; stk_n[0] = (int16_t)n; sp = 1;
    while (sp > 0) {
        sp--;
        int base = stk_s[sp], m = stk_n[sp];
        const uint8_t *s = prot + base;
        if (W > m) continue;
        // Seg::seqent@0x43a2e0
        {
            uint8_t comp[20], sv[21];
            int start = 0;
            mc_seg_comp(s, W, comp);
            mc_seg_state(comp, sv);
            double ent = mc_seg_entropy(T, W, sv);
            for (int i = 0; i <= m - 1; i++) {
                H[i] = ent;
                if (start + 1 + W <= m) {
                    if (s[start] < 20) comp[s[start]]--;
                    if (s[start + W] < 20) comp[s[start + W]]++;
                    start++;
                    mc_seg_state(comp, sv);
                    ent = mc_seg_entropy(T, W, sv);
                }
            }
        }
        int last = m - 1, lowlim = 0;
        for (int i = 0; i <= last; i++) {
            if (H[i] <= 2.2 && H[i] != -1.0) {
                int j, loi, hii, leftend, rightend;
                for (j = i; j >= lowlim; j--) { if (H[j] > 2.5) break; }
                loi = j + 1;
                for (j = i; j <= last; j++) { if (H[j] > 2.5) break; }
                hii = j - 1;
                leftend = loi; rightend = hii;
                mc_seg_trim(T, s + leftend, rightend - leftend + 1, &leftend, &rightend);
                if (i < leftend) {
                    int lend = loi, rend = leftend - 1;
                    if (sp < 16) { stk_s[sp] = (int16_t)(base + lend); stk_n[sp] = (int16_t)(rend - lend + 1); sp++; }
                }
                for (j = leftend; j <= rightend; j++) maskbits[(base + j) >> 3] |= (uint8_t)(1u << ((base + j) & 7));
                i = (hii < rightend) ? hii : rightend;
                lowlim = i + 1;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// SEG, workspace version used by the kernel: identical results to mc_seg_mask, but
//   * the per-window entropies are reduced to the two facts SEG ever uses (H <= 2.2, H <= 2.5), kept as bit sets;
//   * the sorted state vector is maintained incrementally (Seg::decrementsv/incrementsv@0x438d30/0x438d70)
//     instead of being re-sorted for every window;
//   * all arrays live in a caller-provided workspace (LDS on the device): comp[20], sv[24], stack[16] (int16 pairs).
// ---------------------------------------------------------------------------------------------
struct McSegWS { uint8_t *comp; uint8_t *sv; int16_t *stk; };   // 20 B, 24 B, 32 B

MC_HD void mc_sv_dec(uint8_t *sv, int x)
{ // a class with count x loses one member: the LAST entry equal to x becomes x-1 (keeps the vector sorted)
    int i = 0;
    while (sv[i] != 0 && sv[i] != x) i++;
    while (sv[i + 1] == x) i++;
    sv[i] = (uint8_t)(x - 1);
}
MC_HD void mc_sv_inc(uint8_t *sv, int x)
{ // a class with count x (possibly 0 = new class) gains one member: the FIRST entry equal to x becomes x+1
    int i = 0;
    while (sv[i] != x) i++;
    sv[i] = (uint8_t)(x + 1);
    if (x == 0) sv[i + 1] = 0;
}
MC_HD void mc_seg_shift(uint8_t *comp, uint8_t *sv, int out, int in)
{ // Seg::shiftwin1@0x43af30
    if (out < 20) { mc_sv_dec(sv, comp[out]); comp[out]--; }
    if (in < 20) { mc_sv_inc(sv, comp[in]); comp[in]++; }
}
MC_HD void mc_seg_trim_ws(const double *lnfac, const uint8_t *s, int n, int *leftend, int *rightend, const McSegWS &ws)
{
    int lend = 0, rend = n - 1, minlen = 1;
    double minprob = 1.0;
    if (n - 100 > minlen) minlen = n - 100;
    for (int len = n; len > minlen; len--) {
        mc_seg_comp(s, len, ws.comp);
        mc_seg_state(ws.comp, ws.sv);
        for (int i = 0;; i++) {
            double prob = mc_seg_getprob(lnfac, ws.sv, len);
            if (prob < minprob) { minprob = prob; lend = i; rend = len + i - 1; }
            if (i + 1 + len > n) break;
            mc_seg_shift(ws.comp, ws.sv, s[i], s[i + len]);
        }
    }
    *leftend = *leftend + lend;
    *rightend = *rightend - (n - rend - 1);
}
// ---- trimming of a short stretch (<= 15 residues) entirely in registers ---------------------------------------------
// Same search as mc_seg_trim_ws (every window of every length, the least probable one wins, the first on a tie), but the
// composition (20 counts) and the sorted state vector are packed 4-bit fields of two registers each, so that sliding the
// window and evaluating Seg::getprob touch no memory but the ln n! table.  The double arithmetic of getprob is performed
// in the reference's order, so the probabilities are bit-identical.  Low-complexity stretches are short (7.7 residues on
// average for 150 bp reads); longer ones use the workspace version.
struct McRgState { uint64_t clo; uint32_t chi; uint64_t sv; };   // counts of classes 0..15 / 16..19, sorted counts (entry i = nibble i)
#define MC_RG_ONES 0x1111111111111111ull
MC_HD uint64_t mc_rg_eqmask(uint64_t sv, int x)
{ // bit 4i+3 set <=> nibble i of sv equals x (exact, no borrow artefacts)
    const uint64_t y = sv ^ ((uint64_t)x * MC_RG_ONES);
    return ~(((y & 0x7777777777777777ull) + 0x7777777777777777ull) | y) & 0x8888888888888888ull;
}
MC_HD void mc_rg_add(McRgState &st, int r)
{ // Seg::incrementsv: the FIRST entry equal to the class' count grows (a new class appends a 1)
    if (r >= 20) return;
    const int sh = (r & 15) * 4;
    const int x = (r < 16) ? (int)((st.clo >> sh) & 15) : (int)((st.chi >> sh) & 15);
    const uint64_t m = mc_rg_eqmask(st.sv, x);
    st.sv += 1ull << (__builtin_ctzll(m) - 3);
    if (r < 16) st.clo += 1ull << sh; else st.chi += 1u << sh;
}
MC_HD void mc_rg_remove(McRgState &st, int r)
{ // Seg::decrementsv: the LAST entry equal to the class' count shrinks
    if (r >= 20) return;
    const int sh = (r & 15) * 4;
    const int x = (r < 16) ? (int)((st.clo >> sh) & 15) : (int)((st.chi >> sh) & 15);
    const uint64_t m = mc_rg_eqmask(st.sv, x);
    st.sv -= 1ull << (60 - (__builtin_clzll(m) & ~3));
    if (r < 16) st.clo -= 1ull << sh; else st.chi -= 1u << sh;
}
MC_HD double mc_rg_getprob(const double *lnfac, uint64_t sv, int total)
{ // Seg::getprob@0x4393d0 on the packed state vector (at most 15 classes, so the 20-class special case cannot occur)
    double ans1 = lnfac[20];
    int first = (int)(sv & 15);
    if (first != 0) {
        int tot = 20, cls = 1, prev = first;
        uint64_t w = sv >> 4;
        for (;;) {
            const int cur = (int)(w & 15);
            w >>= 4;
            if (cur == prev) cls++;
            else {
                tot -= cls;
                ans1 = ans1 - lnfac[cls];
                if (cur == 0) { ans1 = ans1 - lnfac[tot]; break; }
                cls = 1;
            }
            prev = cur;
        }
    }
    double ans2 = lnfac[total];
    for (uint64_t w = sv; (w & 15) != 0; w >>= 4) ans2 = ans2 - lnfac[w & 15];
    const double t = (double)total * 2.995732273553991;
    return (ans2 + ans1) - t;
}
// ---- Seg::getprob of the register path as a table -------------------------------------------------------------------
// A window of up to 15 residues has one of a few thousand (length, state vector) pairs - the partitions of every t <= len into
// counts - and its probability depends on nothing else.  The host evaluates mc_rg_getprob once per pair (same IEEE double
// operations as the device would perform, same order: bit-identical) and the kernel reads the order-preserving key of the
// result out of an open-addressing table: key word = histogram of the state vector's counts | length << 60.  8192 slots of 16 bytes.
// The key of a pair is not the sorted state vector but the HISTOGRAM of its counts (nibble c - 1 = number of classes that occur
// c times; at most 15 classes, counts up to 15: 60 bits): the same information, and sliding a window changes two nibbles of it
// - no search for "the first entry equal to the class' count" as in Seg::incrementsv / decrementsv.
struct McRhState { uint64_t clo; uint32_t chi; uint64_t hist; };   // counts of classes 0..15 / 16..19 (4 bit each), histogram of the counts
MC_HD void mc_rh_add(McRhState &st, int r)
{
    if (r >= 20) return;
    const int sh = (r & 15) * 4;
    const int x = (r < 16) ? (int)((st.clo >> sh) & 15) : (int)((st.chi >> sh) & 15);       // the class' count so far
    st.hist += 1ull << (4 * x);                                                             // one more class with x + 1
    if (x) st.hist -= 1ull << (4 * (x - 1));                                                // one less with x
    if (r < 16) st.clo += 1ull << sh; else st.chi += 1u << sh;
}
MC_HD void mc_rh_remove(McRhState &st, int r)
{
    if (r >= 20) return;
    const int sh = (r & 15) * 4;
    const int x = (r < 16) ? (int)((st.clo >> sh) & 15) : (int)((st.chi >> sh) & 15);       // >= 1
    st.hist -= 1ull << (4 * (x - 1));
    if (x > 1) st.hist += 1ull << (4 * (x - 2));
    if (r < 16) st.clo -= 1ull << sh; else st.chi -= 1u << sh;
}
MC_HD uint64_t mc_rh_of_sv(uint64_t sv)
{ // histogram of a sorted state vector (host: table build, checks)
    uint64_t h = 0;
    for (uint64_t w = sv; (w & 15) != 0; w >>= 4) h += 1ull << (4 * ((w & 15) - 1));
    return h;
}
#define MC_SEGTAB_LOG2 13
#define MC_SEGTAB_SLOTS (1u << MC_SEGTAB_LOG2)
MC_HD uint64_t mc_seg_prob_key(double x)
{ // unsigned keys that order like the doubles
    uint64_t u;
    __builtin_memcpy(&u, &x, 8);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
// Every pair lies in one of TWO slots (cuckoo placement, mc_build_segtab): a reader fetches both at once and never walks - with
// linear probing one pair in seven sat behind another one, and a round of 64 lanes x 8 windows then waited for a chain of
// dependent loads in nearly every round.
// The key word is sparse (few low bits set in a few nibbles: multiplicative hashes of its halves collide in all 32 bits for dozens
// of pairs), so it is first packed into 32 bits without loss - the histogram's nibbles for counts 1..8 need 19 bits of the low
// word, the seven one-bit nibbles for counts 9..15 and the length go into the gaps - and then mixed by one multiplication.  The
// multiplier was searched for so that the placement exists (the set of pairs is a constant; mc_build_segtab checks).
MC_HD void mc_segtab_slots(uint64_t k, uint32_t &h1, uint32_t &h2)
{
    const uint32_t lo = (uint32_t)k, hi = (uint32_t)(k >> 32);
    uint32_t x = (lo ^ (hi << 7) ^ (((hi >> 28) * 0x924000u) & 0x04444000u)) * 0xBA1BF32Fu;
    x ^= x >> 15;
    h1 = x >> (32 - MC_SEGTAB_LOG2);
    h2 = (x >> 6) & (MC_SEGTAB_SLOTS - 1);
}
// tab: MC_SEGTAB_SLOTS x {key word, probability key}; an empty slot has key word 0 (no pair has an empty state vector AND length 0)
MC_HD uint64_t mc_segtab_lookup(const uint64_t *tab, uint64_t hist, int len)
{
    const uint64_t k = hist | ((uint64_t)len << 60);
    uint32_t h1, h2;
    mc_segtab_slots(k, h1, h2);
    if (tab[2 * h1] == k) return tab[2 * h1 + 1];
    return tab[2 * h2] == k ? tab[2 * h2 + 1] : 0;            // (0 is not reached: every pair a window can have is in the table)
}

MC_HD void mc_seg_trim_rg(const double *lnfac, const uint8_t *s, int n, int *leftend, int *rightend)
{ // n <= 15 (so n - maxtrim < 1: every window length down to 2 is tried)
    int lend = 0, rend = n - 1;
    double minprob = 1.0;
    McRgState st0; st0.clo = 0; st0.chi = 0; st0.sv = 0;
    for (int k = 0; k < n; k++) mc_rg_add(st0, s[k]);
    for (int len = n; len > 1; len--) {
        McRgState st = st0;                                   // window [0, len)
        for (int i = 0;; i++) {
            const double prob = mc_rg_getprob(lnfac, st.sv, len);
            if (prob < minprob) { minprob = prob; lend = i; rend = len + i - 1; }
            if (i + 1 + len > n) break;
            mc_rg_remove(st, s[i]); mc_rg_add(st, s[i + len]);
        }
        mc_rg_remove(st0, s[len - 1]);                        // window [0, len - 1) for the next length
    }
    *leftend = *leftend + lend;
    *rightend = *rightend - (n - rend - 1);
}
struct McBits192 { uint64_t a, b, c; };   // bit set over <= 192 positions kept in registers (no dynamic indexing)
MC_HD void mc_bits_clear(McBits192 &x) { x.a = 0; x.b = 0; x.c = 0; }
MC_HD void mc_bits_set(McBits192 &x, int i) { uint64_t m = 1ull << (i & 63); if (i < 64) x.a |= m; else if (i < 128) x.b |= m; else x.c |= m; }
MC_HD bool mc_bits_test(const McBits192 &x, int i) { uint64_t w = (i < 64) ? x.a : (i < 128) ? x.b : x.c; return (w >> (i & 63)) & 1; }

MC_HDN void mc_seg_mask_ws(const McTables &T, uint8_t *prot, int n, const McSegWS &ws)
{ // masks prot in place (masked residues become MC_INV) - only AFTER all ranges were analysed on the original residues
    int W = (n <= 11) ? 8 : 12;
    McBits192 lo, hi, mk;
    mc_bits_clear(mk);
    if (W > n) return;
    int sp = 1;
    ws.stk[0] = 0; ws.stk[1] = (int16_t)n;
    bool any = false;
    while (sp > 0) {
        sp--;
        int base = ws.stk[2 * sp], m = ws.stk[2 * sp + 1];
        const uint8_t *s = prot + base;
        if (W > m) continue;
        mc_bits_clear(lo); mc_bits_clear(hi);
        {
            int start = 0;
            bool anylo = false;
            mc_seg_comp(s, W, ws.comp);
            mc_seg_state(ws.comp, ws.sv);
            double ent = mc_seg_entropy(T, W, ws.sv);
            for (int i = 0; i <= m - 1; i++) {
                if (ent <= 2.2) { mc_bits_set(lo, i); anylo = true; }
                if (ent <= 2.5) mc_bits_set(hi, i);
                if (start + 1 + W <= m) {
                    mc_seg_shift(ws.comp, ws.sv, s[start], s[start + W]);
                    start++;
                    ent = mc_seg_entropy(T, W, ws.sv);
                }
            }
            if (!anylo) continue;
        }
        int last = m - 1, lowlim = 0;
        for (int i = 0; i <= last; i++) {
            if (mc_bits_test(lo, i)) {
                int j, loi, hii, leftend, rightend;
                for (j = i; j >= lowlim; j--) { if (!mc_bits_test(hi, j)) break; }
                loi = j + 1;
                for (j = i; j <= last; j++) { if (!mc_bits_test(hi, j)) break; }
                hii = j - 1;
                leftend = loi; rightend = hii;
                mc_seg_trim_ws(T.lnfac, s + leftend, rightend - leftend + 1, &leftend, &rightend, ws);
                if (i < leftend) {
                    int lend = loi, rend = leftend - 1;
                    if (sp < 8) { ws.stk[2 * sp] = (int16_t)(base + lend); ws.stk[2 * sp + 1] = (int16_t)(rend - lend + 1); sp++; }
                }
                for (j = leftend; j <= rightend; j++) mc_bits_set(mk, base + j);
                any = true;
                i = (hii < rightend) ? hii : rightend;
                lowlim = i + 1;
            }
        }
    }
    if (any) for (int i = 0; i < n; i++) if (mc_bits_test(mk, i)) prot[i] = MC_INV;
}

// ---------------------------------------------------------------------------------------------
// SEG as the kernel runs it.  Same segmentation as mc_seg_mask_ws, but the entropy of the sliding window is never
// evaluated: with S = sum over residue classes of c*log2(c) and t = number of valid residues in the window,
//     H = log2(t) - S/t,   so   H <= cut   <=>   S >= t*(log2(t) - cut).
// S is kept in fixed point and updated when a residue leaves / enters the window (two table reads), the two cuts become
// two integer compares against per-t constants.  mc_seg_fx_verify (mc_index.h; run by mc_set_run and by the tests) checks
// over every composition a window can have - every partition of every t <= W - that the integer tests decide exactly
// like the reference's double arithmetic (Seg::entropy_cal@0x438f70), including the compositions whose entropy equals a
// cut (e.g. 2,2,1,1,1,1: exactly 2.5).  The rare low-complexity stretches go through the
// same trimming code as before (mc_seg_trim_ws, double precision).  Workspace: comp[20] and the stack; sv is only
// touched by the trimming.
// ---------------------------------------------------------------------------------------------
MC_HDN void mc_seg_mask_fx(const double *lnfac /* McTables::lnfac */, const int32_t *fx /* seg_dout, seg_din, seg_tlo, seg_thi: 4 x 16 */, uint8_t *prot, int n, const McSegWS &ws)
{
    const int W = (n <= 11) ? 8 : 12;
    McBits192 lo, hi, mk;
    mc_bits_clear(mk);
    if (W > n) return;
    int sp = 1;
    ws.stk[0] = 0; ws.stk[1] = (int16_t)n;
    bool any = false;
    while (sp > 0) {
        sp--;
        const int base = ws.stk[2 * sp], m = ws.stk[2 * sp + 1];
        const uint8_t *s = prot + base;
        if (W > m) continue;
        mc_bits_clear(lo); mc_bits_clear(hi);
        {
            bool anylo = false;
            int S = 0, t = 0;
            for (int i = 0; i < 20; i++) ws.comp[i] = 0;
            for (int i = 0; i < W; i++) { const int r = s[i]; if (r < 20) { const int c = ws.comp[r]; S += fx[16 + c]; ws.comp[r] = (uint8_t)(c + 1); t++; } }
            int start = 0;
            bool l = S >= fx[32 + t], h = S >= fx[48 + t];
            for (int i = 0; i <= m - 1; i++) {
                if (l) { mc_bits_set(lo, i); anylo = true; }
                if (h) mc_bits_set(hi, i);
                if (start + 1 + W <= m) {
                    const int o = s[start], e = s[start + W];
                    if (o < 20) { const int c = ws.comp[o]; S += fx[c]; ws.comp[o] = (uint8_t)(c - 1); t--; }
                    if (e < 20) { const int c = ws.comp[e]; S += fx[16 + c]; ws.comp[e] = (uint8_t)(c + 1); t++; }
                    start++;
                    l = S >= fx[32 + t]; h = S >= fx[48 + t];
                }
            }
            if (!anylo) continue;
        }
        int last = m - 1, lowlim = 0;
        for (int i = 0; i <= last; i++) {
            if (mc_bits_test(lo, i)) {
                int j, loi, hii, leftend, rightend;
                for (j = i; j >= lowlim; j--) { if (!mc_bits_test(hi, j)) break; }
                loi = j + 1;
                for (j = i; j <= last; j++) { if (!mc_bits_test(hi, j)) break; }
                hii = j - 1;
                leftend = loi; rightend = hii;
                if (rightend - leftend + 1 <= 15) mc_seg_trim_rg(lnfac, s + leftend, rightend - leftend + 1, &leftend, &rightend);
                else mc_seg_trim_ws(lnfac, s + leftend, rightend - leftend + 1, &leftend, &rightend, ws);
                if (i < leftend) {
                    int lend = loi, rend = leftend - 1;
                    if (sp < 8) { ws.stk[2 * sp] = (int16_t)(base + lend); ws.stk[2 * sp + 1] = (int16_t)(rend - lend + 1); sp++; }
                }
                for (j = leftend; j <= rightend; j++) mc_bits_set(mk, base + j);
                any = true;
                i = (hii < rightend) ? hii : rightend;
                lowlim = i + 1;
            }
        }
    }
    if (any) for (int i = 0; i < n; i++) if (mc_bits_test(mk, i)) prot[i] = MC_INV;
}

// ---- the same masking with the window flags computed ONCE per frame -------------------------------------------------------
// A sub-segment [base, base + m) that Seg::segseq pushes (the part of a stretch left of the trimmed window) is scanned again
// by the reference, but its window k is the frame's window base + k as long as it fits into the sub-segment, and every later
// position reuses the last window that fits: flag_sub(i) = F(base + min(i, m - W)), F(x) = flags of the frame's window that
// starts at x.  So F is computed once (one pass of the incremental fixed-point entropy test) and every segment reads its
// flags off it with shifts; stretches are found with count-trailing-zeros instead of loops.  mc_seg_mask_fx is the plain
// statement; tests/emul checks the two frame by frame.
MC_HD McBits192 mc_bits_shr(const McBits192 &x, int k)
{ // x >> k, 0 <= k < 192
    McBits192 r;
    uint64_t w0 = x.a, w1 = x.b, w2 = x.c;
    if (k >= 128) { w0 = w2; w1 = 0; w2 = 0; k -= 128; }
    else if (k >= 64) { w0 = w1; w1 = w2; w2 = 0; k -= 64; }
    if (k) { r.a = (w0 >> k) | (w1 << (64 - k)); r.b = (w1 >> k) | (w2 << (64 - k)); r.c = w2 >> k; }
    else { r.a = w0; r.b = w1; r.c = w2; }
    return r;
}
MC_HD McBits192 mc_bits_low(int n)
{ // bits [0, n) set, 0 <= n <= 192
    McBits192 r;
    r.a = n >= 64 ? ~0ull : (n > 0 ? (1ull << n) - 1 : 0);
    r.b = n >= 128 ? ~0ull : (n > 64 ? (1ull << (n - 64)) - 1 : 0);
    r.c = n >= 192 ? ~0ull : (n > 128 ? (1ull << (n - 128)) - 1 : 0);
    return r;
}
MC_HD McBits192 mc_bits_and(const McBits192 &x, const McBits192 &y) { McBits192 r; r.a = x.a & y.a; r.b = x.b & y.b; r.c = x.c & y.c; return r; }
MC_HD McBits192 mc_bits_or(const McBits192 &x, const McBits192 &y) { McBits192 r; r.a = x.a | y.a; r.b = x.b | y.b; r.c = x.c | y.c; return r; }
MC_HD McBits192 mc_bits_andnot(const McBits192 &x, const McBits192 &y) { McBits192 r; r.a = x.a & ~y.a; r.b = x.b & ~y.b; r.c = x.c & ~y.c; return r; }
MC_HD McBits192 mc_bits_range(int lo, int hi) { return mc_bits_andnot(mc_bits_low(hi + 1), mc_bits_low(lo)); }   // bits [lo, hi]
// lowest set bit at or above `from` (192 if none)
MC_HD int mc_bits_next(const McBits192 &x, int from)
{
    if (from >= 192) return 192;
    McBits192 y = mc_bits_andnot(x, mc_bits_low(from));
    if (y.a) return __builtin_ctzll(y.a);
    if (y.b) return 64 + __builtin_ctzll(y.b);
    if (y.c) return 128 + __builtin_ctzll(y.c);
    return 192;
}
// highest set bit at or below `from` (-1 if none)
MC_HD int mc_bits_prev(const McBits192 &x, int from)
{
    if (from < 0) return -1;
    McBits192 y = mc_bits_and(x, mc_bits_low(from + 1));
    if (y.c) return 191 - __builtin_clzll(y.c);
    if (y.b) return 127 - __builtin_clzll(y.b);
    if (y.a) return 63 - __builtin_clzll(y.a);
    return -1;
}
// the flags of segment [base, base + m) read off the frame's window flags F (windows 0 .. nF-1 ... nF = n - W + 1)
MC_HD McBits192 mc_seg_flags_of(const McBits192 &F, int base, int m, int W)
{
    McBits192 r = mc_bits_and(mc_bits_shr(F, base), mc_bits_low(m - W + 1));
    if (mc_bits_test(r, m - W)) r = mc_bits_or(r, mc_bits_range(m - W + 1, m - 1));
    return r;
}
// F: the window flags of a frame (one incremental pass; comp: 20 bytes of workspace)
MC_HD void mc_seg_window_flags(const int32_t *fx, const uint8_t *s, int n, int W, uint8_t *comp, McBits192 &Flo, McBits192 &Fhi)
{
    mc_bits_clear(Flo); mc_bits_clear(Fhi);
    int S = 0, t = 0;
    for (int i = 0; i < 20; i++) comp[i] = 0;
    for (int i = 0; i < W; i++) { const int r = s[i]; if (r < 20) { const int c = comp[r]; S += fx[16 + c]; comp[r] = (uint8_t)(c + 1); t++; } }
    for (int x = 0;; x++) {
        if (S >= fx[32 + t]) mc_bits_set(Flo, x);
        if (S >= fx[48 + t]) mc_bits_set(Fhi, x);
        if (x + 1 + W > n) break;
        const int o = s[x], e = s[x + W];
        if (o < 20) { const int c = comp[o]; S += fx[c]; comp[o] = (uint8_t)(c - 1); t--; }
        if (e < 20) { const int c = comp[e]; S += fx[16 + c]; comp[e] = (uint8_t)(c + 1); t++; }
    }
}
// The same pass with the composition in registers (20 counts of 4 bits: a window holds at most 12 residues).  With the counts in
// memory every step is a chain of dependent byte accesses - read a count, look its term up, write it back, and the next read
// must wait for that write; here the loop stores nothing, so the loads of the residues and of the terms run ahead.
MC_HD void mc_seg_window_flags_rg(const int32_t *fx, const uint8_t *s, int n, int W, McBits192 &Flo, McBits192 &Fhi)
{
    uint64_t clo = 0; uint32_t chi = 0;
    int S = 0, t = 0;
    for (int i = 0; i < W; i++) {
        const int r = s[i];
        if (r < 20) {
            const int sh = (r & 15) * 4;
            const int c = (r < 16) ? (int)((clo >> sh) & 15) : (int)((chi >> sh) & 15);
            S += fx[16 + c]; t++;
            if (r < 16) clo += 1ull << sh; else chi += 1u << sh;
        }
    }
    const int nwin = n - W + 1;
    uint64_t wl[3] = {0, 0, 0}, wh[3] = {0, 0, 0};
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int q = 0; q < 3; q++) {
        uint64_t a = 0, b = 0;
        const int x1 = nwin - 64 * q < 64 ? nwin - 64 * q : 64;
        for (int xx = 0; xx < x1; xx++) {
            const int x = 64 * q + xx;
            a |= (uint64_t)(S >= fx[32 + t]) << xx;
            b |= (uint64_t)(S >= fx[48 + t]) << xx;
            if (x + 1 < nwin) {
                const int o = s[x], e = s[x + W];
                if (o < 20) {
                    const int sh = (o & 15) * 4;
                    const int c = (o < 16) ? (int)((clo >> sh) & 15) : (int)((chi >> sh) & 15);
                    S += fx[c]; t--;
                    if (o < 16) clo -= 1ull << sh; else chi -= 1u << sh;
                }
                if (e < 20) {
                    const int sh = (e & 15) * 4;
                    const int c = (e < 16) ? (int)((clo >> sh) & 15) : (int)((chi >> sh) & 15);
                    S += fx[16 + c]; t++;
                    if (e < 16) clo += 1ull << sh; else chi += 1u << sh;
                }
            }
        }
        wl[q] = a; wh[q] = b;
    }
    Flo.a = wl[0]; Flo.b = wl[1]; Flo.c = wl[2];
    Fhi.a = wh[0]; Fhi.b = wh[1]; Fhi.c = wh[2];
}
MC_HDN void mc_seg_mask_fx2(const double *lnfac, const int32_t *fx, uint8_t *prot, int n, const McSegWS &ws)
{
    const int W = (n <= 11) ? 8 : 12;
    if (W > n) return;
    McBits192 Flo, Fhi, mk;
    mc_bits_clear(mk);
    mc_seg_window_flags(fx, prot, n, W, ws.comp, Flo, Fhi);
    if (!(Flo.a | Flo.b | Flo.c)) return;
    int sp = 1;
    ws.stk[0] = 0; ws.stk[1] = (int16_t)n;
    bool any = false;
    while (sp > 0) {
        sp--;
        const int base = ws.stk[2 * sp], m = ws.stk[2 * sp + 1];
        const uint8_t *s = prot + base;
        if (W > m) continue;
        const McBits192 lo = mc_seg_flags_of(Flo, base, m, W), hi = mc_seg_flags_of(Fhi, base, m, W);
        const McBits192 nhi = mc_bits_andnot(mc_bits_low(m), hi);                 // positions of the segment that are NOT high
        int lowlim = 0;
        for (int i = mc_bits_next(lo, 0); i < m; i = mc_bits_next(lo, i + 1)) {
            int loi = mc_bits_prev(nhi, i) + 1; if (loi < lowlim) loi = lowlim;
            int hii = mc_bits_next(nhi, i) - 1; if (hii > m - 1) hii = m - 1;
            int leftend = loi, rightend = hii;
            if (rightend - leftend + 1 <= 15) mc_seg_trim_rg(lnfac, s + leftend, rightend - leftend + 1, &leftend, &rightend);
            else mc_seg_trim_ws(lnfac, s + leftend, rightend - leftend + 1, &leftend, &rightend, ws);
            if (i < leftend) {
                const int lend = loi, rend = leftend - 1;
                if (sp < 8) { ws.stk[2 * sp] = (int16_t)(base + lend); ws.stk[2 * sp + 1] = (int16_t)(rend - lend + 1); sp++; }
            }
            mk = mc_bits_or(mk, mc_bits_range(base + leftend, base + rightend));
            any = true;
            i = (hii < rightend) ? hii : rightend;
            lowlim = i + 1;
        }
    }
    if (any) for (int i = 0; i < n; i++) if (mc_bits_test(mk, i)) prot[i] = MC_INV;
}

// ---------------------------------------------------------------------------------------------
// suffix keys (ExtendSeq2Set 0x413bd2-0x414aa1)
// ---------------------------------------------------------------------------------------------
MC_HD int mc_klen(uint32_t k)
{
    int low = (k & 0xf) != 0xf;
    int l = ((k & 0xff) != 0xff) ? 3 + low : 2 + low;
    l -= ((k & 0xfff) == 0xfff);
    l -= (k == 0xffff);
    return l;
}
MC_HD bool mc_key_lb_less(uint32_t dbk, uint32_t qk)
{
    int ld = mc_klen(dbk), lq = mc_klen(qk), n = ld < lq ? ld : lq;
    if (n != 0) { int sh = (4 - n) * 4; int a = (int)(dbk >> sh), b = (int)(qk >> sh); if (a != b) return a < b; }
    return ld < lq;
}
MC_HD bool mc_key_ub_less(uint32_t qk, uint32_t dbk)
{
    int ld = mc_klen(dbk), lq = mc_klen(qk), n = lq <= ld ? lq : ld;
    if (n == 0) return lq < ld;
    int sh = (4 - n) * 4; int a = (int)(qk >> sh), b = (int)(dbk >> sh);
    if (a == b) return false;
    return a < b;
}
struct McSeedCount { uint32_t lookups, keyprobes, tasks; };   // bucket-bound reads (8 B each) and suffix-key reads (2 B each)

// range of bucket `seed` whose key matches qk on the common prefix; returns ned-nst (0 = nothing)
MC_HD int mc_key_range(const McIndex &X, int seed, uint32_t qk, int *nst_out, McSeedCount *sc)
{
    uint32_t b0 = X.bstart[seed];
    int n = (int)(X.bstart[seed + 1] - b0);
    const uint16_t *keys = X.keys + b0;
    sc->lookups++;
    if (n == 0) return 0;
    int lo = 0, len = n;
    while (len > 0) { int half = len >> 1; sc->keyprobes++; if (mc_key_lb_less(keys[lo + half], qk)) { lo += half + 1; len -= half + 1; } else len = half; }
    int nst = lo;
    if (nst == n) return 0;
    uint32_t dk = keys[nst];
    int m = mc_klen(qk) < mc_klen(dk) ? mc_klen(qk) : mc_klen(dk);
    if (m == 0) return 0;
    int sh = (4 - m) * 4;
    if ((dk >> sh) != (qk >> sh)) return 0;
    lo = 0; len = n;
    while (len > 0) { int half = len >> 1; sc->keyprobes++; if (mc_key_ub_less(qk, keys[lo + half])) len = half; else { lo += half + 1; len -= half + 1; } }
    *nst_out = nst;
    return lo - nst;
}
// ---- sub-bucket records: the fast form of the same lookup -------------------------------------------------------
// A bucket is ordered by CompDbObj, so its postings form contiguous groups by the first reduced residue after the 6-mer
// (postings at the very end of a sequence, key FFFF, first; mc_build_index verifies the grouping).  McBucketRec holds the
// group boundaries: cum[k] = number of postings in front of group k (k = 0..10, 10 = the invalid residue), cum[11] = bucket
// size.  A query key can only match inside the group of its own first residue, so the
// range search runs over that group alone: counting over its first 8 keys when it is that short (89 % of the probes of
// 150 bp reads; 56 % hit an empty group and need no key at all), binary searches otherwise.  Counting equals
// std::lower_bound / std::upper_bound because the group is partitioned with respect to both comparators;
// tests/test_emul.py checks the equality of range, start index and probe count exhaustively on the marker index.
struct McBucketRec { uint32_t start; uint16_t cum[12]; uint16_t pad[2]; };   // 32 B

MC_HD int mc_klen_fast(uint32_t k) { return 4 - (__builtin_ctz((~k & 0xFFFFu) | 0x10000u) >> 2); }

// number of key reads of the reference's binary search (lower_bound or upper_bound) over n keys that ends at index x
MC_HD uint32_t mc_bsearch_reads(int n, int x)
{
    uint32_t c = 0;
    int lo = 0, len = n;
    while (len > 0) { int half = len >> 1; c++; if (lo + half < x) { lo += half + 1; len -= half + 1; } else len = half; }
    return c;
}

// group of 1..8 keys at kp: returns the number of matching postings, *lb_out = index of the first one inside the group
MC_HD int mc_group_range8(const uint16_t *kp, int ns, uint32_t qk, int *lb_out)
{
    // the 8 keys as five aligned 32-bit words (the key array is padded, reading past the group is harmless)
    const uintptr_t a = (uintptr_t)kp;
    const uint32_t *w = (const uint32_t *)(a & ~(uintptr_t)3);
    uint32_t d0 = w[0], d1 = w[1], d2 = w[2], d3 = w[3], d4 = w[4];
    if (a & 2) { d0 = (d0 >> 16) | (d1 << 16); d1 = (d1 >> 16) | (d2 << 16); d2 = (d2 >> 16) | (d3 << 16); d3 = (d3 >> 16) | (d4 << 16); }
    const uint32_t k[8] = {d0 & 0xFFFFu, d0 >> 16, d1 & 0xFFFFu, d1 >> 16, d2 & 0xFFFFu, d2 >> 16, d3 & 0xFFFFu, d3 >> 16};
    const int lq = mc_klen_fast(qk);
    int lb = 0, ub = 0;
    for (int i = 0; i < 8; i++) {
        const uint32_t dk = k[i];
        const int ld = mc_klen_fast(dk), m = ld < lq ? ld : lq, sh = (4 - m) * 4;
        const int x = (int)(dk >> sh), y = (int)(qk >> sh);
        const bool in = i < ns;
        const bool less_db = (m != 0 && x != y) ? (x < y) : (ld < lq);       // mc_key_lb_less(dk, qk)
        const bool less_q = (m == 0) ? (lq < ld) : (y < x);                   // mc_key_ub_less(qk, dk)
        lb += (in && less_db);
        ub += (in && !less_q);
    }
    *lb_out = lb;                                         // where the reference's lower_bound stops, range or not
    if (lb == ns) return 0;
    uint32_t at = k[0];
    for (int i = 1; i < 8; i++) if (i == lb) at = k[i];
    const int la = mc_klen_fast(at), m = lq < la ? lq : la;
    if (m == 0) return 0;
    const int sh = (4 - m) * 4;
    if ((at >> sh) != (qk >> sh)) return 0;
    return ub - lb;
}

// The same for a PROBE of the seed kernel (three or four key residues, then only the 0xF pad): a database key is inside the range
// exactly when its first three / four residues equal the probe's - a shorter key differs at its own pad and sorts in front of
// the range, a longer one compares on the probe's length (mc_key_lb_less / mc_key_ub_less).  *lb_out is set for a range only.
// Checked against mc_group_range8 over every key of the index and its near misses (tests/emul, MC_CHECK_SCAN).
MC_HD int mc_group_match8(const uint16_t *kp, int ns, uint32_t qk, int *lb_out)
{
    const uintptr_t a = (uintptr_t)kp;
    const uint32_t *w = (const uint32_t *)(a & ~(uintptr_t)3);
    uint32_t d0 = w[0], d1 = w[1], d2 = w[2], d3 = w[3], d4 = w[4];
    if (a & 2) { d0 = (d0 >> 16) | (d1 << 16); d1 = (d1 >> 16) | (d2 << 16); d2 = (d2 >> 16) | (d3 << 16); d3 = (d3 >> 16) | (d4 << 16); }
    const int sh = (4 - mc_klen_fast(qk)) * 4;
    const uint32_t q2 = (qk >> sh) * 0x10001u, lm = (0xFFFFu >> sh) * 0x10001u;
    const uint32_t x0 = ((d0 >> sh) & lm) ^ q2, x1 = ((d1 >> sh) & lm) ^ q2, x2 = ((d2 >> sh) & lm) ^ q2, x3 = ((d3 >> sh) & lm) ^ q2;
    uint32_t m = ((x0 & 0xFFFFu) == 0 ? 1u : 0u) | ((x0 >> 16) == 0 ? 2u : 0u) | ((x1 & 0xFFFFu) == 0 ? 4u : 0u) | ((x1 >> 16) == 0 ? 8u : 0u) |
                 ((x2 & 0xFFFFu) == 0 ? 16u : 0u) | ((x2 >> 16) == 0 ? 32u : 0u) | ((x3 & 0xFFFFu) == 0 ? 64u : 0u) | ((x3 >> 16) == 0 ? 128u : 0u);
    m &= (1u << ns) - 1u;
    *lb_out = __builtin_ctz(m | 256u);
    return __builtin_popcount(m);
}

// the same for a group of any length, by the reference's two binary searches
MC_HD int mc_group_range_bs(const uint16_t *kp, int ns, uint32_t qk, int *lb_out)
{
    int lo = 0, len = ns;
    while (len > 0) { int half = len >> 1; if (mc_key_lb_less(kp[lo + half], qk)) { lo += half + 1; len -= half + 1; } else len = half; }
    const int lb = lo;
    *lb_out = lb;
    if (lb == ns) return 0;
    const uint32_t at = kp[lb];
    const int lq = mc_klen_fast(qk), la = mc_klen_fast(at), m = lq < la ? lq : la;
    if (m == 0) return 0;
    const int sh = (4 - m) * 4;
    if ((at >> sh) != (qk >> sh)) return 0;
    lo = 0; len = ns;
    while (len > 0) { int half = len >> 1; if (mc_key_ub_less(qk, kp[lo + half])) len = half; else { lo += half + 1; len -= half + 1; } }
    return lo - lb;
}

// record-based equivalent of mc_key_range (same result, same start index, same algorithmic probe counts)
MC_HD int mc_key_range_rec(const McBucketRec *rec, const uint16_t *keys, int seed, uint32_t qk, int *nst_out, McSeedCount *sc)
{
    const McBucketRec *R = rec + seed;
    const int k6 = (int)(qk >> 12);                       // 0..9, or 10 = the invalid group
    const int c0 = R->cum[k6], ns = (int)R->cum[k6 + 1] - c0, n = R->cum[11];
    sc->lookups++;
    if (n == 0) return 0;
    int lb = 0, cnt = 0;
    if (ns > 0) cnt = ns <= 8 ? mc_group_range8(keys + R->start + c0, ns, qk, &lb) : mc_group_range_bs(keys + R->start + c0, ns, qk, &lb);
    // key reads of the reference: its lower_bound over the whole bucket stops at c0 + lb; upper_bound runs only for a range
    sc->keyprobes += mc_bsearch_reads(n, c0 + lb) + (cnt > 0 ? mc_bsearch_reads(n, c0 + lb + cnt) : 0u);
    *nst_out = c0 + lb;
    return cnt;
}
// ---- 9-mer filter -------------------------------------------------------------------------------------------------
// An exact 9-mer probe (3-residue key, written g6 g7 g8 F) matches exactly the postings whose first three key residues
// equal its own (keys shorter than 3 sort in front of the range: mc_key_lb_less, equal prefix, shorter first).  Most exact
// probes of a read find nothing, so the seed kernel asks a Bloom filter first: it holds (bucket, k | 0xF) of every posting
// with at least 3 key residues, 2 bits in one 32-bit word, 2^18 words (1 MB, L2 resident).  No false negatives; a positive
// goes through the exact range search.  (The one-substitution 10-mer probes have the wildcard and pair filters below.)
#ifndef MC_FILT9_LOG2W
#define MC_FILT9_LOG2W 18
#endif
#define MC_FILT9_WORDS (1u << MC_FILT9_LOG2W)
MC_HD uint32_t mc_filter_hash(uint32_t bucket, uint32_t key)
{
    uint32_t x = bucket * 0x9E3779B1u + key * 0x85EBCA77u;
    x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 13;
    return x;
}
MC_HD uint32_t mc_filter9_word(uint32_t h) { return h >> (32 - MC_FILT9_LOG2W); }
MC_HD uint32_t mc_filter_bits(uint32_t h) { return (1u << (h & 31)) | (1u << ((h >> 5) & 31)); }

// ---- wildcard filter: one probe instead of ten ----------------------------------------------------------------------
// The neighbourhood of a position is its 10-mer with ONE residue substituted, at offset 3, 4 or 5 (a digit of the bucket)
// or 6 (the first key residue): 4 groups of 9 probes.  Instead of asking about every probe, the kernel first asks, per
// group, "does the index hold ANY 10-mer that equals mine except at this offset?".  The index 10-mers are entered four
// times, each time with one of the four middle residues left out.  Layout for locality: the six outer residues (offsets
// 0-2 and 7-9) pick a 32-byte line, the wildcard position picks one of its four 64-bit parts, the three remaining middle
// residues pick one bit in each of the part's two words - so the four questions of a position cost ONE 32-byte read and
// three operations each.  A negative answer is exact (no 10-mer of the group can match); a positive one sends the group
// to the pair filter.  2^19 lines = 16 MB (rounds 2 - 4: 2^18 - 92 pair filter asks per read of 150 bp, now 77, and those blocks
// are cache misses all: seed kernel 6.29 -> 6.16 ms per 1 M reads; 2^20 lines: 6.14).
#ifndef MC_WILD_LOG2L
#define MC_WILD_LOG2L 19
#endif
#define MC_WILD_LINES (1u << MC_WILD_LOG2L)
#define MC_WILD_LINE_WORDS 8
MC_HD uint32_t mc_wild_ctx(uint32_t seed, uint32_t key) { return (seed / 1000u) * 4096u + (key & 0xFFFu); }
MC_HD uint32_t mc_wild_line(uint32_t ctx)
{
    uint32_t x = ctx * 0x9E3779B1u;
    x ^= x >> 15; x *= 0x85EBCA77u; x ^= x >> 13;
    return x >> (32 - MC_WILD_LOG2L);
}
// group g: 0 = offset 4 (stride 10), 1 = offset 5 (stride 1), 2 = offset 3 (stride 100), 3 = offset 6 (first key residue)
// The two bit positions of a group hash the line context and the three middle residues that are NOT the wildcard: one sum
// over all four (mc_wild_sum), minus the wildcard residue's own term, then one mixing round per group.
#define MC_WILD_K3 0x297A2D39u
#define MC_WILD_K4 0x68E31DA5u
#define MC_WILD_K5 0x1B56C4E9u
#define MC_WILD_K6 0x4CF5AD43u
MC_HD uint32_t mc_wild_sum(uint32_t ctx, uint32_t r3, uint32_t r4, uint32_t r5, uint32_t r6) { return ctx * 0x2C1B3C6Du + r3 * MC_WILD_K3 + r4 * MC_WILD_K4 + r5 * MC_WILD_K5 + r6 * MC_WILD_K6; }
MC_HD uint32_t mc_wild_bits_s(uint32_t sum, uint32_t rg, int g)
{ // rg: the residue at the group's wildcard offset; result: a bit position in each of the part's two words (low and high byte)
    uint32_t x = sum - rg * (g == 0 ? MC_WILD_K4 : g == 1 ? MC_WILD_K5 : g == 2 ? MC_WILD_K3 : MC_WILD_K6) + 0x51ED27u * (uint32_t)(g + 1);
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15;
    return (x & 31u) | (((x >> 5) & 31u) << 8);
}
MC_HD uint32_t mc_wild_bits(uint32_t ctx, uint32_t seed, uint32_t key, int g)
{
    const uint32_t r3 = (seed / 100u) % 10u, r4 = (seed / 10u) % 10u, r5 = seed % 10u, r6 = key >> 12;
    return mc_wild_bits_s(mc_wild_sum(ctx, r3, r4, r5, r6), g == 0 ? r4 : g == 1 ? r5 : g == 2 ? r3 : r6, g);
}
MC_HD bool mc_wild_test2(uint32_t w0, uint32_t w1, uint32_t bits) { return (((w0 >> (bits & 31u)) & (w1 >> (bits >> 8))) & 1u) != 0; }
MC_HD bool mc_wild_test(const uint32_t q[2], uint32_t bits) { return mc_wild_test2(q[0], q[1], bits); }
MC_HD void mc_wild_set(uint32_t q[2], uint32_t bits) { q[0] |= 1u << (bits & 31u); q[1] |= 1u << (bits >> 8); }

// ---- pair filter: the ten probes of a (position, wildcard offset) pair in ONE 16-byte read ------------------------------
// A pair that passed the wildcard filter used to ask the 10-mer Bloom filter once per substituted residue: nine scattered
// words, nine L2 requests - and the seed kernel is bound by the requests its CU can keep in flight.  Here the nine known
// residues and the offset pick a 128-bit block; the block holds ten 12-bit cells, one per residue value at the wildcard
// offset, and the context picks three of the twelve bits (the same three in every cell).  An index 10-mer sets its three bits
// in the cell of its own residue, once per offset; a query reads the block and gets the mask of residues that may complete
// an index 10-mer with a handful of 64-bit operations.  No false negatives; false positives go through the exact search.
#ifndef MC_PAIR_LOG2B
#define MC_PAIR_LOG2B 20
#endif
#define MC_PAIR_BLOCKS (1u << MC_PAIR_LOG2B)
// group g: 0 = offset 4 (bucket digit of stride 10), 1 = offset 5 (stride 1), 2 = offset 3 (stride 100), 3 = offset 6 (first key residue)
MC_HD uint32_t mc_pair_digit(uint32_t seed, uint32_t key, int g) { return g == 0 ? (seed / 10u) % 10u : g == 1 ? seed % 10u : g == 2 ? (seed / 100u) % 10u : key >> 12; }
MC_HD uint32_t mc_pair_hash_d(uint32_t seed, uint32_t key, int g, uint32_t d)
{ // hash of the 10-mer with the residue at the wildcard offset (d) taken out
    const uint32_t st = g == 0 ? 10u : g == 1 ? 1u : g == 2 ? 100u : 0u;
    const uint32_t s0 = seed - d * st, k0 = g == 3 ? (key & 0x0FFFu) : key;
    uint32_t x = s0 * 0x9E3779B1u + k0 * 0x85EBCA77u + (uint32_t)(g + 1) * 0x51ED270Bu;
    x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 13; x *= 0x297A2D39u; x ^= x >> 16;
    return x;
}
MC_HD uint32_t mc_pair_hash(uint32_t seed, uint32_t key, int g) { return mc_pair_hash_d(seed, key, g, mc_pair_digit(seed, key, g)); }
MC_HD uint32_t mc_pair_block(uint32_t h) { return h >> (32 - MC_PAIR_LOG2B); }
MC_HD uint32_t mc_pair_mix(uint32_t h) { uint32_t y = h * 0x9E3779B1u; return y ^ (y >> 15); }   // the block index uses the top bits of h: the pattern gets bits of its own
MC_HD uint32_t mc_pair_bit_a(uint32_t y) { return ((y & 0xFFu) * 12u) >> 8; }
MC_HD uint32_t mc_pair_bit_b(uint32_t y) { return (((y >> 8) & 0xFFu) * 12u) >> 8; }
MC_HD uint32_t mc_pair_bit_c(uint32_t y) { return (((y >> 16) & 0xFFu) * 12u) >> 8; }
MC_HD void mc_pair_set(uint32_t q[4], uint32_t h, uint32_t j)
{ // cells 0..4 in words 0,1 (bit 12 j), cells 5..9 in words 2,3
    if (j > 9) return;
    const uint32_t y = mc_pair_mix(h);
    const uint32_t base = (j < 5 ? 0u : 64u) + 12u * (j < 5 ? j : j - 5u), pa = base + mc_pair_bit_a(y), pb = base + mc_pair_bit_b(y), pc = base + mc_pair_bit_c(y);
    q[pa >> 5] |= 1u << (pa & 31); q[pb >> 5] |= 1u << (pb & 31); q[pc >> 5] |= 1u << (pc & 31);
}
MC_HD uint32_t mc_pair_test4(uint32_t x, uint32_t y, uint32_t z, uint32_t w, uint32_t h)
{ // bit j of the result: residue j at the wildcard offset may complete an index 10-mer
    const uint32_t m = mc_pair_mix(h), a = mc_pair_bit_a(m), b = mc_pair_bit_b(m), c = mc_pair_bit_c(m);
    const unsigned long long lo = (unsigned long long)x | ((unsigned long long)y << 32), hi = (unsigned long long)z | ((unsigned long long)w << 32);
    const unsigned long long ml = (lo >> a) & (lo >> b) & (lo >> c), mh = (hi >> a) & (hi >> b) & (hi >> c);
    uint32_t r = 0;
    for (int j = 0; j < 5; j++) r |= ((uint32_t)(ml >> (12 * j)) & 1u) << j | ((uint32_t)(mh >> (12 * j)) & 1u) << (j + 5);
    return r;
}

// ---- range table: the answer for probes into long groups -----------------------------------------------------------
// A probe whose first-residue group holds more than 8 keys needs the reference's two binary searches: ~20 dependent
// loads, and such probes are the rule for true hits (a conserved 10-mer occurs in hundreds of homologous markers).  The
// kernel does not search for them: every (bucket, key) that has a non-empty range inside a long group is entered, with
// the range mc_key_range itself returns, into an open-addressing hash table (8-byte slots: bucket 20 | key 16 | start 11
// | count 11 bits; 0xFFFF... = empty).  A probe that is not in the table has no range.  (The counting form of the kernel
// still searches: it has to report where the reference's lower_bound stops even when there is no range.)
MC_HD uint32_t mc_rt_hash(uint32_t bucket, uint32_t key)
{
    uint32_t x = bucket * 0x85EBCA77u + key * 0x9E3779B1u;
    x ^= x >> 16; x *= 0x2C1B3C6Du; x ^= x >> 15;
    return x;
}
MC_HD unsigned long long mc_rt_pack(uint32_t bucket, uint32_t key, uint32_t nst, uint32_t cnt)
{
    return ((unsigned long long)bucket << 38) | ((unsigned long long)key << 22) | ((unsigned long long)nst << 11) | (unsigned long long)cnt;
}
// returns the number of postings (0 = none), *nst_out = index of the first one inside the bucket
MC_HD int mc_rt_lookup(const unsigned long long *rt, uint32_t mask, uint32_t bucket, uint32_t key, int *nst_out)
{
    const unsigned long long tag = ((unsigned long long)bucket << 16) | key;
    for (uint32_t i = mc_rt_hash(bucket, key) & mask;; i = (i + 1) & mask) {
        const unsigned long long e = rt[i];
        if ((e >> 22) == tag) { *nst_out = (int)((e >> 11) & 0x7FF); return (int)(e & 0x7FF); }
        if (e == ~0ull) return 0;
    }
}

MC_HD uint32_t mc_pack_key(const uint8_t *g, int nkey)
{
    uint32_t qk = 0;
    for (int i = 0; i < nkey; i++) qk |= (uint32_t)g[i] << (12 - 4 * i);
    for (int i = nkey; i <= 3; i++) qk |= 0xfu << (12 - 4 * i);
    return qk & 0xffff;
}

// ---------------------------------------------------------------------------------------------
// seed enumeration for one frame (Searching@0x415050) - sequential over positions because of `prev`.
// emit(seed_bucket, nst, count, seedlen, nkey, pos, phase) is called for every non-empty posting range.
// phase: 0 exact; 1..30 = 1 + stride_index*10 + j; 31..40 = 31 + substituted first key residue.
// ---------------------------------------------------------------------------------------------
template <class Emit>
MC_HDN void mc_enumerate_seeds(const McTables &T, const McIndex &X, const uint8_t *q, int qlen, Emit &emit, McSeedCount *sc)
{
    if (qlen <= 6) return;
    int prev = 6;
    for (int pos = 0; pos + 6 < qlen; pos++) {
        int seed = 0, len = 6, used, g;
        bool bad = false;
        for (int k = 0; k < 6; k++) { g = T.grp[q[pos + k]]; if (g == MC_INVGRP) { bad = true; break; } seed = seed * 10 + g; }
        if (bad) continue;
        uint32_t freq = X.bstart[seed + 1] - X.bstart[seed];
        sc->lookups++;
        if (freq > T.freq_thr) {
            int rest = qlen - pos - 6, maxextra = (rest >= 2) ? 3 : rest + 1;
            if (maxextra <= 1) len = 7;
            else {
                double thr = (double)T.freq_thr;
                int extra = 1, idx = pos + 7;
                g = T.grp[q[pos + 6]];
                if (g == MC_INVGRP) continue;
                double e = (double)freq * T.letter_p[g];
                if (!(thr >= e)) {
                    for (;;) {
                        extra++;
                        if (!(maxextra > extra)) break;
                        g = T.grp[q[idx]];
                        if (g == MC_INVGRP) { bad = true; break; }
                        idx++;
                        e = e * T.letter_p[g];
                        if (thr >= e) break;
                    }
                    if (bad) continue;
                }
                len = 6 + extra;
            }
        }
        used = (len >= prev - 1) ? len : prev - 1;
        if (qlen < pos + used) continue;
        uint8_t key[4];
        for (int k = 6; k < used && k < 10; k++) key[k - 6] = T.grp[q[pos + k]];
        if (freq != 0) {
            int r = 0, nst = 0;
            if (used > 6) r = mc_key_range(X, seed, mc_pack_key(key, used - 6), &nst, sc);
            else r = (int)freq;
            if (r > 0) emit(seed, nst, r, used, used - 6, pos, 0);
            prev = used;
            if (r <= 0) prev = 6;
        }
        if (!(qlen < pos + 10)) {
            bool ok = true;
            for (int k = pos + used; k < pos + 10; k++) if (T.grp[q[k]] == MC_INVGRP) { ok = false; break; }
            if (!ok) continue;
            for (int k = 0; k < 4; k++) key[k] = T.grp[q[pos + 6 + k]];
            uint32_t qk = mc_pack_key(key, 4);
            const int strides[3] = {10, 1, 100};
            for (int m = 0; m < 3; m++) {
                int st = strides[m], start = seed - ((seed / st) % 10) * st;
                for (int j = 0; j < 10; j++) {
                    int v = start + j * st, nst = 0;
                    if (v == seed) continue;
                    int r = mc_key_range(X, v, qk, &nst, sc);
                    if (r > 0) emit(v, nst, r, 10, 4, pos, 1 + m * 10 + j);
                }
            }
            int orig = key[0];
            for (int k = 0; k < 10; k++) {
                if (k == orig) continue;
                key[0] = (uint8_t)k;
                int nst = 0, r = mc_key_range(X, seed, mc_pack_key(key, 4), &nst, sc);
                if (r > 0) emit(seed, nst, r, 10, 4, pos, 31 + k);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// one seed hit: redundancy skip, seed score, growth, gate, ungapped X-drop (ExtendSeq2Set
// 0x413e68-0x41422a + AlignSeqs 0x413370-0x4137a7).  Result: 0 = dropped, 1 = ungapped HSP complete
// (out), 2 = needs gapped extension (gt).
// ---------------------------------------------------------------------------------------------
#define MC_SUB(T, a, b) ((int)(T).sub[((a) << 5) | (b)])

// TT: McTables, or a compact copy of its hot members (sub, grp, xdrop_*, gap_trigger) - the kernels keep one in LDS
// QP / DP: pointers to the frame and to the subject's residues (d[i] = residue i of the marker) - plain memory, or the
// kernel's LDS copies: the frame, and the subject window [max(0, dpos - qpos), min(dlen, dpos + qlen - qpos)), which is all this function can touch
template <class TT, class QP, class DP>
MC_HDN int mc_eval_seed_tail(const TT &T, QP q, int qlen, int frame, int qpos, DP d, int dlen, int dpos, int sidx,
                             int seedlen, int score, int ident, McGapTask *gt);
template <class TT, class QP, class DP>
MC_HDN int mc_eval_seed_core(const TT &T, QP q, int qlen, int frame, int qpos, DP d, int dlen, int dpos, int sidx,
                             int seedlen, int nkey, McGapTask *gt)
{
    if (dpos + seedlen > dlen) return 0;
    if (qpos != 0 && dpos != 0 && T.grp[q[qpos - 1]] == T.grp[d[dpos - 1]] && nkey != 4) return 0;
    int score = 0, ident = 0;
    for (int k = 0; k < seedlen; k++) { int a = q[qpos + k], b = d[dpos + k]; score += MC_SUB(T, a, b); ident += (a == b); }
    return mc_eval_seed_tail(T, q, qlen, frame, qpos, d, dlen, dpos, sidx, seedlen, score, ident, gt);
}
// ... from the seed's own score on: growth in both directions, the gate, the ungapped X-drop extension
template <class TT, class QP, class DP>
MC_HDN int mc_eval_seed_tail(const TT &T, QP q, int qlen, int frame, int qpos, DP d, int dlen, int dpos, int sidx,
                             int seedlen, int score, int ident, McGapTask *gt)
{
    int L = seedlen;
    int lim = dlen - dpos; if (lim > qlen - qpos) lim = qlen - qpos;
    while (lim > L && T.grp[q[qpos + L]] == T.grp[d[dpos + L]]) { int a = q[qpos + L], b = d[dpos + L]; score += MC_SUB(T, a, b); ident += (a == b); L++; }
    int back = qpos < dpos ? qpos : dpos, qp = qpos, dp = dpos;
    while (back > 0 && T.grp[q[qp - 1]] == T.grp[d[dp - 1]]) { qp--; dp--; back--; L++; int a = q[qp], b = d[dp]; score += MC_SUB(T, a, b); ident += (a == b); }
    if (!((double)score >= MC_SEED_SCORE && ident >= MC_SEED_IDENT)) return 0;
    int s0 = score, qfwd = 0, qbwd = 0, fgain = 0, bgain = 0;
    { // forward
        int n1 = qlen - qp - L, n2 = dlen - dp - L, bl = 0, bi = 0;
        if (n1 != 0 && n2 != 0 && !(s0 < -20)) {
            QP p1 = q + qp + L;
            DP p2 = d + dp + L;
            int run = s0, best = s0, id = 0;
            for (int i = 0;;) {
                int a = p1[i], b = p2[i];
                run += MC_SUB(T, a, b); id += (a == b); i++;
                if (run > best) { best = run; bl = i; bi = id; }
                if (!(n2 > i)) break;
                if (n1 <= i) break;
                if (run < -20) break;
                if ((double)run < (double)best - T.xdrop_ungapped) break;
            }
            fgain = best - s0;
        }
        ident += bi; qfwd = bl;
    }
    { // backward, restarting from the seed score
        int a = qp - 1, b = dp - 1, bl = 0, bi = 0;
        if (a >= 0 && b >= 0 && !(s0 < -20)) {
            int run = s0, best = s0, id = 0, cnt = 0;
            for (;;) {
                int x = q[a], y = d[b];
                run += MC_SUB(T, x, y); id += (x == y); cnt++;
                if (best < run) { best = run; bl = cnt; bi = id; }
                a--; b--;
                if (b < 0) break;
                if (a < 0) break;
                if (run < -20) break;
                if ((double)run < (double)best - T.xdrop_ungapped) break;
            }
            bgain = best - s0;
        }
        ident += bi; qbwd = bl;
    }
    score = s0 + bgain + fgain;
    gt->sidx = (uint32_t)sidx; gt->qp = (int16_t)qp; gt->dp = (int16_t)dp; gt->L = (int16_t)L;
    gt->qfwd = (int16_t)qfwd; gt->qbwd = (int16_t)qbwd; gt->score = (int16_t)score; gt->nmatch = (int16_t)ident;
    (void)frame;
    return (!(T.gap_trigger > (double)score)) ? 2 : 1;
}
template <class TT>
MC_HDN int mc_eval_seed(const TT &T, const McIndex &X, const uint8_t *q, int qlen, int frame, int qpos,
                        uint32_t posting, int seedlen, int nkey, McGapTask *gt)
{
    const int dpos = (int)(posting & 0x7ff), sidx = (int)(posting >> 11);
    const uint32_t o0 = X.off[sidx];
    return mc_eval_seed_core(T, q, qlen, frame, qpos, X.res + o0, (int)(X.off[sidx + 1] - o0), dpos, sidx, seedlen, nkey, gt);
}

// ---------------------------------------------------------------------------------------------
// Gapped X-drop extension without trace planes (AlignGapped@0x40a550).
// The reference records three trace planes and walks back from the best cell; the only things it keeps
// from that walk are: residues consumed, identities, the number of gap runs and of gap columns, and the
// number of trace steps.  Those are carried forward here, per DP state, along exactly the predecessor
// the trace would follow (same tie rules: 's' over E/e over D/d, open over extend on ties in the band,
// strict '>' in the right-growth loop), packed as ident | steps<<10 | runs<<20 | gapcols<<26 ... (64 bit).
// seq1/seq2 are addressed through (base, stride) so the left flank is walked backwards in place.
// ---------------------------------------------------------------------------------------------
struct McPath { uint16_t ident, steps, gapcols; uint8_t runs, cls; };   // cls: class of the last step 0 s, 1 E, 2 D, 3 none
MC_HD McPath mc_path_zero() { McPath p; p.ident = 0; p.steps = 0; p.gapcols = 0; p.runs = 0; p.cls = 3; return p; }
MC_HD McPath mc_path_gap(McPath p, int cls)
{ // append one gap step of class cls (1 = E: consumes seq2, 2 = D: consumes seq1)
    if (p.cls != cls) p.runs++;
    p.cls = (uint8_t)cls; p.steps++; p.gapcols++;
    return p;
}

struct McGapResult { int gain, c1, c2, ident, steps, runs, gapcols, overflow; };

// workspace: H, D (int) and PH, PD (McPath), each n2+2 entries.
//   PH[j] = step sequence (origin .. cell) the trace would follow from main[i][j]
//   PD[j] = the same for the D plane of the cell
// DP state of one subject column: main and D plane scores with the path statistics they carry
struct McGapCell { int H, D; McPath PH, PD; };   // 24 B

// C: workspace of `cap` columns.  The band normally stays within a few columns of the diagonal (max column 53 for 150 bp
// reads), but nothing bounds it below the subject length: when a column >= cap would be written the function gives up
// with R.overflow = 1 and the caller repeats the flank with a full-size workspace.
template <class TT>
MC_HDN McGapResult mc_align_gapped(const TT &T, const uint8_t *s1, int st1, const uint8_t *s2, int st2, int n1, int n2, McGapCell *C, int cap)
{
    const int open = MC_GAP_OPEN, ext = MC_GAP_EXT, first = MC_GAP_OPEN + MC_GAP_EXT;
    McGapResult R; R.overflow = 0; R.gain = 0; R.c1 = 0; R.c2 = 0; R.ident = 0; R.steps = 0; R.runs = 0; R.gapcols = 0;
    int jEnd = (int)((T.xdrop_gapped - (double)open) / (double)ext);   // 0x40a693-0x40a6b9
    int best = 0, bestI = 0, bestJ = 0, jStart = 1;
    McPath bestP = mc_path_zero();
    C[0].H = 0; C[0].D = -open; C[0].PH = mc_path_zero(); C[0].PD = mc_path_zero();
    if (n2 > 0 && jEnd > 0) {                                          // row 0: 'E' then 'e' (0x40a6c1-0x40a770)
        int r = -open;
        McPath pe = mc_path_zero();
        for (int j = 1;;) {
            if (j >= cap) { R.overflow = 1; return R; }
            r -= ext; C[j].H = r; C[j].D = r - open;
            pe = mc_path_gap(pe, 1);
            C[j].PH = pe; C[j].PD = pe;
            j++;
            if (jEnd < j) break;
            if (n2 < j) break;
        }
    }
    if (n1 <= 0 || jEnd <= 1) return R;
    for (int i = 1;;) {
        // left border (i, jStart-1): the planes say 'D' in row 1 (continue from main[0][.]) and 'd' afterwards
        int diag = C[jStart - 1].H;
        McPath pdiag = C[jStart - 1].PH;
        int hprev = C[jStart - 1].H - first;
        if (hprev < C[jStart - 1].D - ext) hprev = C[jStart - 1].D - ext;
        McPath pborder = mc_path_gap((i == 1) ? C[jStart - 1].PH : C[jStart - 1].PD, 2);
        C[jStart - 1].D = hprev; C[jStart - 1].H = hprev;
        C[jStart - 1].PD = pborder; C[jStart - 1].PH = pborder;
        int E = hprev - open, h = 0;
        McPath pE = pborder;       // E plane of the previous cell of this row
        McPath phprev = pborder;   // main path of the previous cell of this row
        bool grow = true, trim = true;
        if (!(jStart > jEnd) && !(n2 < jStart)) {
            for (int j = jStart;;) {                                   // 0x40a8e4-0x40a9f5
                if (j >= cap) { R.overflow = 1; return R; }
                int a = hprev - first, b = E - ext, Dn;
                McPath npE, npD;
                if (a >= b) { E = a; npE = mc_path_gap(phprev, 1); }   // 'E': from main[i][j-1]
                else { E = b; npE = mc_path_gap(pE, 1); }              // 'e': from the E plane of (i, j-1)
                a = C[j].H - first; b = C[j].D - ext;
                if (a >= b) { Dn = a; npD = mc_path_gap(C[j].PH, 2); }   // 'D': from main[i-1][j]
                else { Dn = b; npD = mc_path_gap(C[j].PD, 2); }          // 'd': from the D plane of (i-1, j)
                int x = s1[(i - 1) * st1], y = s2[(j - 1) * st2];
                int s = diag + MC_SUB(T, x, y);
                McPath np = pdiag; np.steps++; np.ident += (x == y); np.cls = 0;
                h = s;
                if (E > h) { h = E; np = npE; }
                if (h < Dn) { h = Dn; np = npD; }
                diag = C[j].H; pdiag = C[j].PH;
                C[j].H = h; C[j].D = Dn; C[j].PH = np; C[j].PD = npD;
                pE = npE; phprev = np; hprev = h;
                if (h > best) { best = h; bestI = i; bestJ = j; bestP = np; }
                else if ((double)best - T.xdrop_gapped > (double)h && j > bestJ) {
                    if (j >= jEnd) { jEnd = j; }                       // 0x40b48f: fall into the right growth
                    else { jEnd = j; grow = false; trim = false; }     // 0x40a9f5: next row
                    break;
                }
                j++;
                if (n2 < j) break;
                if (j > jEnd) break;
            }
        }
        if (grow) {                                                    // 0x40ac45-0x40ad08
            for (int j = jEnd + 1; !(n2 < j); j++) {
                if (j >= cap) { R.overflow = 1; return R; }
                int a = hprev - first, b = E - ext;
                McPath npE;
                if (a > b) { E = a; npE = mc_path_gap(phprev, 1); }
                else { E = b; npE = mc_path_gap(pE, 1); }
                C[j].H = E; C[j].D = E - open; C[j].PH = npE; C[j].PD = npE;
                pE = npE; phprev = npE;
                if (E > best) { best = E; bestI = i; bestJ = j; bestP = npE; }
                else if ((double)best - T.xdrop_gapped > (double)E) { jEnd = j; break; }
                hprev = E;
            }
        }
        if (trim && !(jStart > bestJ)) {                               // 0x40ad0c-0x40ad82
            double lim = (double)best - T.xdrop_gapped;
            if (lim > (double)C[bestJ].H) jStart = bestJ;
            else { int k = bestJ; for (;;) { k--; if (jStart > k) break; if (lim > (double)C[k].H) { jStart = k; break; } } }
        }
        i++;
        if (n1 < i) break;
        if (!(jStart < jEnd)) break;
    }
    R.gain = best; R.c1 = bestI; R.c2 = bestJ;
    if (best > 0) { R.ident = bestP.ident; R.steps = bestP.steps; R.runs = bestP.runs; R.gapcols = bestP.gapcols; }
    return R;
}

// ---------------------------------------------------------------------------------------------
// The same extension on a SLIDING WINDOW of W columns with 12-byte cells - the form the gapped kernels run with the window in LDS.
//
// * Only the columns jStart-1 .. jEnd of the previous row are ever read again (jStart never decreases, a row writes every
//   column it leaves behind up to its new jEnd), so the DP rows live in a circular buffer indexed by column mod W; a write to a
//   column >= (jStart-1) + W means the band is wider than the window: overflow, the caller repeats the flank with a wider window
//   or with mc_align_gapped on a full-size workspace.
// * Path statistics in 24 bits: ident | diag << 8 | runs << 16 (diag = number of substitution steps).  A path from the origin
//   to cell (i, j) has i = diag + D-gap columns and j = diag + E-gap columns, so the gap columns and the number of steps need
//   not be carried: gapcols = i + j - 2 diag, steps = i + j - diag.  No field can overflow: ident <= diag <= MC_MAXAA = 170 <
//   256; every gap run costs >= open + ext = 12 and every substitution step gains <= 11, and a path that is still alive scores
//   >= -xdrop, so runs <= (11 * 170 + 27) / 12 = 158 < 256.
// * The class of a path's last step (mc_align_gapped's McPath::cls, which decides whether a gap step opens a new run) is not
//   carried: it follows from the step itself.  OPENING a gap from a main cell (a >= b) means main - first >= plane - ext, i.e.
//   main >= plane + 11 > plane: the main cell did not take its value from that plane, so its path does not end in a gap of that
//   class - always a new run (the left border of a row and the cells of row 0 / of the right growth have main - D = open or
//   main = D: they tie or lose, see below).  EXTENDING continues the plane's own path: never a new run - except through the D
//   plane of a cell written by row 0 or by the right growth, whose "D path" is the cell's E path (PD = PH there): only the left
//   border of a later row can continue such a cell (an in-band cell over it has main - first = D - ext: a tie, which opens), so
//   those cells carry a flag (MC_PD_GROW) that the border looks at.
// * Scores in 16 bits: |score| <= 11 * 170 + 64.
// * The subject residue of a column travels with the column's cell (read from memory once, when the column enters the band):
//   bits 24..28 of the PD word.
// WS: load(slot, H, D, PH, PD) / store(slot, H, D, PH, PD) / loadH(slot), slot = 0 .. W-1 (the accessor owns the layout).
//
// The extension is a state machine - mc_gap_begin (row 0), mc_gap_row (one DP row), mc_gap_result - so that a GPU lane can take
// the next flank the moment its own ends (k_gapped_lds); mc_align_gapped_win runs it to the end for one flank.
// ---------------------------------------------------------------------------------------------
#define MC_PP_RUN 0x10000u
#define MC_PD_GROW 0x80000000u
#define MC_PD_YMASK 0x1F000000u
#define MC_PD_PATH 0x00FFFFFFu

// A cell in 64 bits (what the LDS windows of k_gapped_lds hold):
//   word 0: H 12 bits (signed) | min(H - D, 11) 4 bits | ident, diag of PH 16 bits
//   word 1: ident, diag of PD 16 bits | runs of PH 5 bits | runs of PD 5 bits | subject residue 5 bits | MC_PD_GROW 1 bit
// D itself is not needed: a cell is only ever asked whether H - first >= D - ext (H - D >= 11: the gap opens, and D drops out of the
// result) and otherwise for D - ext with H - D <= 10 - so H - D is kept, capped at 11, and unpacking returns H - 11 for a capped D
// (same decision, same value).  Returns nonzero when a run count does not fit its 5 bits (a live path with 32 gap runs).
MC_HD uint32_t mc_gap_pack(int H, int D, uint32_t PH, uint32_t PD, uint32_t &w0, uint32_t &w1)
{
    int delta = H - D;
    if (delta > 11) delta = 11;
    w0 = ((uint32_t)H & 0xFFFu) | ((uint32_t)delta << 12) | (PH << 16);
    w1 = (PD & 0xFFFFu) | ((PH >> 16) & 0x1Fu) << 16 | ((PD >> 16) & 0x1Fu) << 21 | ((PD >> 24) & 0x1Fu) << 26 | (PD & MC_PD_GROW);
    return ((PH | PD) >> 21) & 7u;                                  // bits 21..23 of either path word: runs >= 32
}
MC_HD void mc_gap_unpack(uint32_t w0, uint32_t w1, int &H, int &D, uint32_t &PH, uint32_t &PD)
{
    H = (int)(w0 << 20) >> 20;
    D = H - (int)((w0 >> 12) & 15u);
    PH = (w0 >> 16) | ((w1 >> 16) & 0x1Fu) << 16;
    PD = (w1 & 0xFFFFu) | ((w1 >> 21) & 0x1Fu) << 16 | ((w1 >> 26) & 0x1Fu) << 24 | (w1 & MC_PD_GROW);
}

struct McGapState {
    const uint8_t *s1, *s2;
    int st, n1, n2;
    int i, jStart, jEnd, best, bestI, bestJ, base, cbase, x, over;
    uint32_t bestP;
};

// row 0 ('E' then 'e', 0x40a6c1-0x40a770); false: no DP row follows (nothing to extend, or the window is too narrow: S.over).
// have: the subject residues of columns 1 .. 16 were fetched ahead (lo: columns 1 .. 8, hi: 9 .. 16, one byte each) - a GPU lane
// loads them while it is still busy with its previous flank; without them row 0 reads the subject itself.  x0: the first query residue.
template <class TT, class WS>
MC_HD bool mc_gap_begin(const TT &T, McGapState &S, const uint8_t *s1, const uint8_t *s2, int st, int n1, int n2, WS &ws, int W, bool have = false, uint64_t lo = 0, uint64_t hi = 0, int x0 = 0)
{
    const int open = MC_GAP_OPEN, ext = MC_GAP_EXT;
    S.s1 = s1; S.s2 = s2; S.st = st; S.n1 = n1; S.n2 = n2; S.over = 0;
    S.jEnd = (int)((T.xdrop_gapped - (double)open) / (double)ext);
    S.best = 0; S.bestI = 0; S.bestJ = 0; S.jStart = 1; S.bestP = 0; S.base = 0; S.cbase = 0; S.i = 1; S.x = 0;
    ws.store(0, 0, -open, 0u, 0u);
    if (n2 > 0 && S.jEnd > 0) {
        int r = -open;
        const int last = S.jEnd < n2 ? S.jEnd : n2;
        if (last >= W) { S.over = 1; return false; }
        for (int j = 1; j <= last; j++) {
            r -= ext;
            const uint32_t y = (have && j <= 16) ? (uint32_t)((j <= 8 ? lo : hi) >> (8 * ((j - 1) & 7))) & 31u : (uint32_t)s2[(j - 1) * st];
            ws.store(j, r, r - open, MC_PP_RUN, MC_PP_RUN | (y << 24) | MC_PD_GROW);   // one E run from the origin
        }
    }
    if (n1 <= 0 || S.jEnd <= 1) return false;
    S.x = have ? x0 : (int)s1[0];
    return true;
}

// one DP row (0x40a7e0-0x40ad82); true: the extension has ended (S.over: the band left the window)
template <class TT, class WS>
MC_HD bool mc_gap_row(const TT &T, McGapState &S, WS &ws, int W)
{
    const int open = MC_GAP_OPEN, ext = MC_GAP_EXT, first = MC_GAP_OPEN + MC_GAP_EXT;
    const int xd = (int)T.xdrop_gapped;        // (double)best - xdrop > (double)h  <=>  best - h > (int)xdrop  (integers against a positive threshold)
    const int i = S.i, n2 = S.n2, st = S.st;
    int base = S.base, cbase = S.cbase;        // lowest live column and its slot; slot of column j >= base: cbase + (j - base), minus W when >= W
#define MC_WSLOT(j) ((cbase + ((j) - base)) >= W ? (cbase + ((j) - base)) - W : (cbase + ((j) - base)))
    int jStart = S.jStart, jEnd = S.jEnd, best = S.best, bestI = S.bestI, bestJ = S.bestJ;
    uint32_t bestP = S.bestP;
    { const int nb = jStart - 1; cbase = MC_WSLOT(nb); base = nb; }
    const int x = S.x;
    const int xn = (i < S.n1) ? S.s1[i * st] : 0;                     // next row's query residue: its load overlaps this row
    // the subject residues of the first two columns the right growth would add (it starts behind the jEnd of the row's entry, or not
    // at all): in flight during the band
    const uint32_t gy0 = (jEnd + 1 <= n2) ? (uint32_t)S.s2[jEnd * st] : 0u, gy1 = (jEnd + 2 <= n2) ? (uint32_t)S.s2[(jEnd + 1) * st] : 0u;
    const int gfirst = jEnd + 1;
    const int8_t *subrow = T.sub + (x << 5);
    int bH, bD; uint32_t bPH, bPD;
    ws.load(cbase, bH, bD, bPH, bPD);
    int cj = (cbase + 1 == W) ? 0 : cbase + 1;
    int cH, cD; uint32_t cPH, cPD;
    ws.load(cj, cH, cD, cPH, cPD);                                     // the first cell of the band (column jStart): in flight with the border's
    // left border (i, jStart-1): the planes say 'D' in row 1 (continue from main[0][.]) and 'd' afterwards
    int diag = bH;
    uint32_t pdiag = bPH;
    int hprev = bH - first;
    if (hprev < bD - ext) hprev = bD - ext;
    const uint32_t pborder = (i == 1) ? bPH + MC_PP_RUN : (bPD & MC_PD_PATH) + ((bPD & MC_PD_GROW) ? MC_PP_RUN : 0u);
    ws.store(cbase, hprev, hprev, pborder, pborder | (bPD & MC_PD_YMASK));
    int E = hprev - open;
    uint32_t pE = pborder, phprev = pborder;
    bool grow = true, trim = true;
    {   // the band (0x40a8e4-0x40a9f5): columns jStart .. min(jEnd, n2) unless the X-drop test ends the row earlier (on entry jStart < jEnd and
        // jStart <= n2: at least one cell).  One exit test per cell; what the exit means is sorted out behind the loop.  The cell of the
        // NEXT column and its substitution score are fetched while the current one is computed (one slot past the band is read and not used).
        int jLast = jEnd < n2 ? jEnd : n2;
        const bool wide = jLast > base + W - 1;                        // the row would leave the window
        if (wide) jLast = base + W - 1;
        int j = jStart;
        int y = (int)((cPD >> 24) & 31u), sub = (int)subrow[y];
        bool brk;
        for (;;) {
            const int cjn = (cj + 1 == W) ? 0 : cj + 1;
            int nH, nD; uint32_t nPH, nPD;
            ws.load(cjn, nH, nD, nPH, nPD);
            int a = hprev - first, b = E - ext;
            const uint32_t npE = (a >= b) ? phprev + MC_PP_RUN : pE;   // 'E': from main[i][j-1]; 'e': from the E plane of (i, j-1)
            E = a >= b ? a : b;
            a = cH - first; b = cD - ext;
            const uint32_t npD = (a >= b) ? cPH + MC_PP_RUN : (cPD & MC_PD_PATH);   // 'D': from main[i-1][j]; 'd': from the D plane of (i-1, j)
            const int Dn = a >= b ? a : b;
            const int s = diag + sub;
            uint32_t np = pdiag + 0x100u + (x == y ? 1u : 0u);
            int h = s;
            if (E > h) { h = E; np = npE; }                            // ties: s over E, (s | E) over D
            if (h < Dn) { h = Dn; np = npD; }
            diag = cH; pdiag = cPH;
            ws.store(cj, h, Dn, np, npD | (cPD & MC_PD_YMASK));
            pE = npE; phprev = np; hprev = h;
            const bool up = h > best;
            brk = !up && best - h > xd && j > bestJ;
            if (up) { best = h; bestI = i; bestJ = j; bestP = np; }
            const int yn = (int)((nPD >> 24) & 31u);
            const int subn = (int)subrow[yn];
            if (brk || j >= jLast) break;
            j++; cj = cjn; cH = nH; cD = nD; cPH = nPH; cPD = nPD; y = yn; sub = subn;
        }
        if (brk) {
            if (j < jEnd) { grow = false; trim = false; }              // 0x40a9f5: next row; else 0x40b48f: fall into the right growth
            jEnd = j;
        } else if (wide) { S.over = 1; return true; }
    }
    if (grow) {                                                        // 0x40ac45-0x40ad08
        for (int j = jEnd + 1; !(n2 < j); j++) {
            if (j - base >= W) { S.over = 1; return true; }
            const int a = hprev - first, b = E - ext;
            const uint32_t npE = (a > b) ? phprev + MC_PP_RUN : pE;
            E = a > b ? a : b;
            const uint32_t yy = j == gfirst ? gy0 : j == gfirst + 1 ? gy1 : (uint32_t)S.s2[(j - 1) * st];
            ws.store(MC_WSLOT(j), E, E - open, npE, npE | (yy << 24) | MC_PD_GROW);
            pE = npE; phprev = npE;
            if (E > best) { best = E; bestI = i; bestJ = j; bestP = npE; }
            else if (best - E > xd) { jEnd = j; break; }
            hprev = E;
        }
    }
    if (trim && !(jStart > bestJ)) {                                   // 0x40ad0c-0x40ad82
        if (best - ws.loadH(MC_WSLOT(bestJ)) > xd) jStart = bestJ;
        else { int k = bestJ; for (;;) { k--; if (jStart > k) break; if (best - ws.loadH(MC_WSLOT(k)) > xd) { jStart = k; break; } } }
    }
#undef MC_WSLOT
    S.base = base; S.cbase = cbase; S.jStart = jStart; S.jEnd = jEnd; S.best = best; S.bestI = bestI; S.bestJ = bestJ; S.bestP = bestP;
    S.i = i + 1; S.x = xn;
    return S.n1 < S.i || !(jStart < jEnd);
}

MC_HD McGapResult mc_gap_result(const McGapState &S)
{
    McGapResult R; R.overflow = S.over; R.gain = 0; R.c1 = 0; R.c2 = 0; R.ident = 0; R.steps = 0; R.runs = 0; R.gapcols = 0;
    if (S.over) return R;
    R.gain = S.best; R.c1 = S.bestI; R.c2 = S.bestJ;
    if (S.best > 0) {
        const int diagn = (int)((S.bestP >> 8) & 0xFF);
        R.ident = (int)(S.bestP & 0xFF); R.runs = (int)((S.bestP >> 16) & 0xFF);
        R.gapcols = S.bestI + S.bestJ - 2 * diagn; R.steps = S.bestI + S.bestJ - diagn;
    }
    return R;
}

template <class TT, class WS>
MC_HDN McGapResult mc_align_gapped_win(const TT &T, const uint8_t *s1, int st1, const uint8_t *s2, int st2, int n1, int n2, WS &ws, int W)
{
    (void)st2;                                                          // (both sequences are walked in the same direction)
    McGapState S;
    if (mc_gap_begin(T, S, s1, s2, st1, n1, n2, ws, W)) while (!mc_gap_row(T, S, ws, W)) { }
    return mc_gap_result(S);
}

// finalise one HSP (CalRes@0x4077a0 up to the keep test).  Returns false when the HSP is not kept.
MC_HD bool mc_make_hsp(const McTables &T, int ntlen, int frame, const McGapTask &g, int qfwd, int dfwd, int qbwd, int dbwd,
                       int score, int nmatch, int alnlen, int gapopens, int gaptotal, McHsp *h)
{
    if (score < 0 || score >= MC_SMAX) return false;
    double le = T.loge_r[score];
    if (!(score > 30) && !(T.loge_thr > le)) return false;
    int qbeg = g.qp, dbeg = g.dp, L = g.L;
    h->sidx = (int32_t)g.sidx; h->score = (int16_t)score; h->frame = (int16_t)frame; h->loge = le;
    h->alnlen = (int16_t)alnlen; h->gaps = (int16_t)gapopens; h->mism = (int16_t)(alnlen - nmatch - gaptotal); h->nmatch = (int16_t)nmatch;
    h->qaas = (int16_t)(qbeg - qbwd); h->qaae = (int16_t)(qbeg + qfwd - 1 + L);
    h->ds = (int16_t)(dbeg - dbwd); h->de = (int16_t)(dbeg + dfwd - 1 + L);
    if (frame <= 2) { h->qnts = (int16_t)((qbeg - qbwd) * 3 + frame + 1); h->qnte = (int16_t)((qbeg + qfwd + L) * 3 + frame); }
    else { int s = ntlen - (qbeg - qbwd) * 3 - (frame - 3); h->qnts = (int16_t)s; h->qnte = (int16_t)(s + 1 - (qfwd + qbwd + L) * 3); }
    return true;
}
