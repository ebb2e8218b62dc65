/* oracle/rapsearch_port.c - CPU restatement of RAPsearch2 v2.15 (TEST INFRASTRUCTURE ONLY).
 *
 * See rapsearch_port.h.  Addresses in comments are virtual addresses inside
 * /root/reference/microbe_census/bin/rapsearch_Linux_2.15 (objdump -d -C -M intel).
 * The reference launches that binary from microbe_census.py:369-389 with
 *   -z T -e 1 -t n -p f -b 0      (defaults: -v 500, -g t, -a f, -w f)
 * and this port implements exactly that mode.
 *
 * Build:  gcc -O2 -fPIC -shared -o librapsearch_port.so rapsearch_port.c -lm
 *         gcc -O2 -DRS_MAIN -fopenmp -o rs_port rapsearch_port.c -lm
 */
#include "rapsearch_port.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * Residue coding and scoring tables   (CHashSearch::CHashSearch@0x4166d0, 0x4169bd-0x416a62)
 * ---------------------------------------------------------------------------------------- */
#define NBUCKET 1000000
#define INVALID_CODE 0xA0
#define INVALID_GRP 10

static uint8_t g_code[256];       /* char -> code = (group<<4)|(1+index in group); default 0xA0 */
static uint8_t g_grp[256];        /* code -> murphy10 group, 10 = invalid */
static int32_t g_sub[256 * 256];  /* code x code substitution scores, default -5 */
static int g_tables_ready = 0;

static const int8_t B62[20][20] = { /* order ARNDCQEGHILKMFPSTWYV */
    {4, -1, -2, -2, 0, -1, -1, 0, -2, -1, -1, -1, -1, -2, -1, 1, 0, -3, -2, 0},
    {-1, 5, 0, -2, -3, 1, 0, -2, 0, -3, -2, 2, -1, -3, -2, -1, -1, -3, -2, -3},
    {-2, 0, 6, 1, -3, 0, 0, 0, 1, -3, -3, 0, -2, -3, -2, 1, 0, -4, -2, -3},
    {-2, -2, 1, 6, -3, 0, 2, -1, -1, -3, -4, -1, -3, -3, -1, 0, -1, -4, -3, -3},
    {0, -3, -3, -3, 9, -3, -4, -3, -3, -1, -1, -3, -1, -2, -3, -1, -1, -2, -2, -1},
    {-1, 1, 0, 0, -3, 5, 2, -2, 0, -3, -2, 1, 0, -3, -1, 0, -1, -2, -1, -2},
    {-1, 0, 0, 2, -4, 2, 5, -2, 0, -3, -3, 1, -2, -3, -1, 0, -1, -3, -2, -2},
    {0, -2, 0, -1, -3, -2, -2, 6, -2, -4, -4, -2, -3, -3, -2, 0, -2, -2, -3, -3},
    {-2, 0, 1, -1, -3, 0, 0, -2, 8, -3, -3, -1, -2, -1, -2, -1, -2, -2, 2, -3},
    {-1, -3, -3, -3, -1, -3, -3, -4, -3, 4, 2, -3, 1, 0, -3, -2, -1, -3, -1, 3},
    {-1, -2, -3, -4, -1, -2, -3, -4, -3, 2, 4, -2, 2, 0, -3, -2, -1, -2, -1, 1},
    {-1, 2, 0, -1, -3, 1, 1, -2, -1, -3, -2, 5, -1, -3, -1, 0, -1, -3, -2, -2},
    {-1, -1, -2, -3, -1, 0, -2, -3, -2, 1, 2, -1, 5, 0, -2, -1, -1, -1, -1, 1},
    {-2, -3, -3, -3, -2, -3, -3, -3, -1, 0, 0, -3, 0, 6, -4, -2, -2, 1, 3, -1},
    {-1, -2, -2, -1, -3, -1, -1, -2, -2, -3, -3, -1, -2, -4, 7, -1, -1, -4, -3, -2},
    {1, -1, 1, 0, -1, 0, 0, 0, -1, -2, -2, 0, -1, -2, -1, 4, 1, -3, -2, -2},
    {0, -1, 0, -1, -1, -1, -1, -2, -2, -1, -1, -1, -1, -2, -1, 1, 5, -2, -2, 0},
    {-3, -3, -4, -4, -2, -2, -3, -2, -2, -3, -2, -3, -1, 1, -4, -3, -2, 11, 2, -3},
    {-2, -2, -2, -3, -2, -1, -2, -3, 2, -1, -1, -2, -1, 3, -3, -2, -2, 2, 7, -1},
    {0, -3, -3, -3, -1, -2, -2, -3, -3, 3, 1, -2, 1, -1, -2, -2, 0, -3, -1, 4}};

static void init_tables(void)
{
    static const char *groups[10] = {"A", "KR", "EDNQ", "C", "G", "H", "ILVM", "FYW", "P", "S/T"};
    static const char *order = "ARNDCQEGHILKMFPSTWYV";
    int g, k, i, j;
    if (g_tables_ready) return;
    memset(g_code, INVALID_CODE, sizeof g_code);
    memset(g_grp, INVALID_GRP, sizeof g_grp);
    for (g = 0; g < 10; g++)
        for (k = 0; groups[g][k]; k++) {
            uint8_t c = (uint8_t)((g << 4) + 1 + k);
            unsigned char ch = (unsigned char)groups[g][k];
            g_code[ch] = c;
            g_code[ch + 32] = c;
            g_grp[c] = (uint8_t)g;
        }
    for (i = 0; i < 256 * 256; i++) g_sub[i] = -5;
    for (i = 0; i < 20; i++)
        for (j = 0; j < 20; j++)
            g_sub[(g_code[(unsigned char)order[i]] << 8) | g_code[(unsigned char)order[j]]] = B62[i][j];
    g_tables_ready = 1;
}

#define SUB(a, b) g_sub[((a) << 8) | (b)]

/* thresholds (CHashSearch::Search 0x418cee-0x418ddc; BlastStat::Bits2RawScore*@0x437d90/0x437db0) */
#define LN2 0.6931471805599453
static const double T_GAPTRIG = (25.0 * LN2 - 2.0099154790312257) / 0.318; /* this+0x40378 */
static const double T_XUNGAP = (7.0 * LN2 - 2.0099154790312257) / 0.318;   /* this+0x40388 */
static const double T_XGAP = (15.0 * LN2 - 3.1941832122778293) / 0.267;    /* this+0x40398 */
#define T_SEEDSCORE 11.0 /* this+0x403a0 */
#define T_SEEDIDENT 4    /* this+0x403a8 */
#define GAP_OPEN 11      /* this+0x40358 */
#define GAP_EXT 1        /* this+0x4035c */
static double LOGE_THR = 1.0; /* -e 1 (this+0x40350); RS_LOGE_THR overrides it for differential tests */
#define MAX_M8 500       /* -v 500 (this+0x40430) */

/* ------------------------------------------------------------------------------------------
 * Database  (as serialised by prerapsearch: BuildDHash@0x40fc20; loaded in Search@0x418750)
 * ---------------------------------------------------------------------------------------- */
struct rs_db {
    int64_t nres;
    uint8_t *res;       /* residue codes of all sequences, concatenated */
    int nseq;
    uint32_t *off;      /* nseq+1 offsets */
    char **names;
    int64_t *bstart;    /* NBUCKET+1 */
    uint32_t *post;     /* posting = seqIdx<<11 | pos */
    uint16_t *keys;     /* 4 reduced residues after the 6-mer, 4 bit each, 0xF past the end */
    int64_t npost;
    uint32_t freq_thr;  /* median of all bucket sizes as stored in .info (0 for the marker DB) */
    double letter_p[10];
    /* statistics (BlastStat::BlastStat@0x438820, SetDBInfo@0x4387b0) */
    int adj_table[1000];
};

static uint64_t rd_u64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
static uint32_t rd_u32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }

static uint8_t *slurp(const char *path, int64_t *n)
{
    FILE *f = fopen(path, "rb");
    uint8_t *b;
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    *n = ftell(f);
    fseek(f, 0, SEEK_SET);
    b = (uint8_t *)malloc((size_t)*n + 16);
    if (fread(b, 1, (size_t)*n, f) != (size_t)*n) { free(b); fclose(f); return NULL; }
    fclose(f);
    return b;
}

static int length_adjustment(const rs_db *db, int m);

rs_db *rs_db_load_rapdb(const char *path)
{
    int64_t n, p, i;
    char info_path[4096];
    uint8_t *b = slurp(path, &n), *inf;
    rs_db *db;
    if (!b) return NULL;
    init_tables();
    if (getenv("RS_LOGE_THR")) LOGE_THR = atof(getenv("RS_LOGE_THR"));
    db = (rs_db *)calloc(1, sizeof *db);
    p = 0x28; /* 8+22 signature, u16 version, 4 size bytes, u32 1 */
    db->nres = (int64_t)rd_u64(b + p); p += 8;
    db->res = (uint8_t *)malloc((size_t)db->nres);
    memcpy(db->res, b + p, (size_t)db->nres); p += db->nres;
    db->nseq = (int)rd_u64(b + p) - 1; p += 8;
    db->off = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(db->nseq + 1));
    memcpy(db->off, b + p, 4 * (size_t)(db->nseq + 1)); p += 4 * (int64_t)(db->nseq + 1);
    /* vector<vector<uint>>: 5 bytes class info, u64 count, u32 item version */
    p += 5;
    if (rd_u64(b + p) != NBUCKET) { fprintf(stderr, "rapdb: unexpected bucket count\n"); return NULL; }
    p += 8 + 4;
    db->bstart = (int64_t *)malloc(sizeof(int64_t) * (NBUCKET + 1));
    {
        int64_t q = p, tot = 0;
        for (i = 0; i < NBUCKET; i++) { uint64_t c = rd_u64(b + q); q += 8 + 4 * (int64_t)c; tot += (int64_t)c; }
        db->npost = tot;
        db->post = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)tot);
        db->keys = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)tot);
    }
    db->bstart[0] = 0;
    for (i = 0; i < NBUCKET; i++) {
        uint64_t c = rd_u64(b + p); p += 8;
        memcpy(db->post + db->bstart[i], b + p, 4 * (size_t)c); p += 4 * (int64_t)c;
        db->bstart[i + 1] = db->bstart[i] + (int64_t)c;
    }
    p += 5;
    if ((int)rd_u64(b + p) != db->nseq) { fprintf(stderr, "rapdb: name count mismatch\n"); return NULL; }
    p += 8 + 4;
    db->names = (char **)malloc(sizeof(char *) * (size_t)db->nseq);
    for (i = 0; i < db->nseq; i++) {
        uint64_t l = rd_u64(b + p); p += 8;
        db->names[i] = (char *)malloc((size_t)l + 1);
        memcpy(db->names[i], b + p, (size_t)l); db->names[i][l] = 0; p += (int64_t)l;
    }
    p += 5; p += 8 + 4;
    for (i = 0; i < NBUCKET; i++) {
        uint64_t c = rd_u64(b + p); p += 8;
        memcpy(db->keys + db->bstart[i], b + p, 2 * (size_t)c); p += 2 * (int64_t)c;
    }
    free(b);
    /* .info: ... u64 1e6 + counts (byte 60), u32 threshold, u64 10 + 10 doubles */
    snprintf(info_path, sizeof info_path, "%s.info", path);
    inf = slurp(info_path, &n);
    if (inf) {
        int64_t q = 68 + 4 * (int64_t)NBUCKET;
        db->freq_thr = rd_u32(inf + q); q += 4 + 8;
        memcpy(db->letter_p, inf + q, 80);
        free(inf);
    }
    for (i = 0; i < 1000; i++) db->adj_table[i] = (i <= 10) ? 0 : length_adjustment(db, (int)i);
    return db;
}

void rs_db_free(rs_db *db)
{
    int i;
    if (!db) return;
    for (i = 0; i < db->nseq; i++) free(db->names[i]);
    free(db->names); free(db->res); free(db->off); free(db->bstart); free(db->post); free(db->keys);
    free(db);
}
int rs_db_nseq(const rs_db *db) { return db->nseq; }
const char *rs_db_name(const rs_db *db, int s) { return db->names[s]; }
int rs_db_seqlen(const rs_db *db, int s) { return (int)(db->off[s + 1] - db->off[s]); }
const uint8_t *rs_db_residues(const rs_db *db, int64_t *n) { *n = db->nres; return db->res; }
const uint32_t *rs_db_offsets(const rs_db *db) { return db->off; }
const int64_t *rs_db_bucket_starts(const rs_db *db) { return db->bstart; }
const uint32_t *rs_db_postings(const rs_db *db, int64_t *n) { *n = db->npost; return db->post; }
const uint16_t *rs_db_keys(const rs_db *db) { return db->keys; }

/* ------------------------------------------------------------------------------------------
 * Karlin-Altschul statistics, gapped parameter set   (BlastStat::SetPar(1)@0x437f20)
 * ---------------------------------------------------------------------------------------- */
#define KA_LAMBDA 0.267
#define KA_K 0.041
#define KA_ALPHA_D_LAMBDA 7.116105 /* bit pattern 0x401c76e43aa79bbb, checked in init */
#define KA_BETA (-30.0)
#define KA_DECAY 0.1

typedef struct { double ell, mprime, nprime; } ka_eff; /* BlastStat +0x78, +0x28, +0x10 */

/* BlastStat::blastComputeLengthAdjustment@0x438410 (= NCBI BLAST_ComputeLengthAdjustment) */
static int length_adjustment(const rs_db *db, int qlen)
{
    double m = (double)qlen, n = (double)db->nres, nseq = (double)db->nseq;
    double logK = log(KA_K), alpha_d_lambda, beta = KA_BETA;
    double ell, ell_min = 0.0, ell_max, ell_next, ss, ell_bar, mb, c, mx;
    int i, converged = 0, adj;
    union { uint64_t u; double d; } ad; ad.u = 0x401c76e43aa79bbbULL; alpha_d_lambda = ad.d;
    mx = (m > n) ? m : n;              /* maxsd xmm0=m, xmm1=n : returns n unless m > n */
    c = m * n - mx / KA_K;
    if (c < 0) return 0;
    mb = m * nseq + n;
    ell_max = 2 * c / (mb + sqrt(mb * mb + (-4.0) * nseq * c));
    ell_next = 0.0;
    for (i = 1; i <= 20; i++) {
        ell = ell_next;
        ss = (m - ell) * (n - nseq * ell);
        ell_bar = alpha_d_lambda * (log(ss) + logK) + beta;
        if (ell_bar >= ell) {
            ell_min = ell;
            if (ell_bar - ell_min <= 1.0) { converged = 1; break; }
            if (ell_min == ell_max) break;
        } else {
            ell_max = ell;
        }
        if (ell_min <= ell_bar && ell_bar <= ell_max) ell_next = ell_bar;
        else ell_next = (i == 1) ? ell_max : (ell_min + ell_max) * 0.5;
    }
    adj = (int)ell_min;
    if (converged) {
        ell = ceil(ell_min);
        if (ell <= ell_max) {
            ss = (m - ell) * (n - nseq * ell);
            if (alpha_d_lambda * (log(ss) + logK) + beta >= ell) adj = (int)ell;
        }
    }
    return adj;
}

/* blastComputeLengthAdjustmentComp@0x438730 + the state it leaves in BlastStat */
static ka_eff ka_effective(const rs_db *db, int qlen_aa)
{
    ka_eff e;
    int adj = (qlen_aa < 1000) ? db->adj_table[qlen_aa < 0 ? 0 : qlen_aa] : length_adjustment(db, qlen_aa);
    e.ell = (double)adj;
    e.mprime = (double)qlen_aa - e.ell;
    e.nprime = (double)db->nres - (double)db->nseq * e.ell;
    return e;
}

/* BlastStat::rawScore2ExpectLog@0x438270 */
static double ka_loge(const ka_eff *e, double s)
{
    double t = KA_K * e->nprime;
    double x;
    t = t * e->mprime;
    x = exp(s * (-KA_LAMBDA)) * t;
    x = x / (1.0 - KA_DECAY);
    if (x == 0.0) return -10000.0;
    return log(x) / 2.302585092994046;
}
/* BlastStat::rawScore2Bit@0x437d70 */
static double ka_bits(double s) { return (s * KA_LAMBDA - log(KA_K)) / LN2; }

static double fac_i(int n) { int r = 1; while (n > 1) { r *= n; n--; } return (double)r; } /* BlastStat::fac@0x437dd0 (int arithmetic) */

/* BlastStat::sumScore2Expect(int,double*,int)@0x438300 -> (int,double,int)@0x437f90 */
static double ka_sum_expect(const ka_eff *e, int n, const double *scores, int subj_len)
{
    double sum = 0.0, t, a, b, xsum, ex, pw, d, r, x;
    int i;
    for (i = 0; i < n; i++) sum += scores[i];
    a = 1.0 / KA_K;
    b = (double)subj_len - e->ell;
    if (!(a > b)) a = b; /* maxsd xmm3(a), xmm2(b) */
    t = log(KA_K * e->mprime * a);
    xsum = sum * KA_LAMBDA - t;
    xsum = xsum - (double)(n - 1) * (log(KA_K) + 7.824046010856292);
    xsum = xsum - log(fac_i(n));
    ex = exp(-xsum);
    pw = pow(xsum, (double)(n - 1));
    d = pow(0.1, (double)(n - 1)) * 0.9;
    r = e->nprime / (double)subj_len;
    x = ex * pw;
    x = x / (fac_i(n) * fac_i(n - 1));
    x = x / d;
    return r * x;
}

/* ------------------------------------------------------------------------------------------
 * SEG low-complexity masking (Seg::*@0x438b50-0x43b140; NCBI seg with window W, K1 2.2, K2 2.5,
 * maxtrim 100, and - as Seg::initialize@0x439650 sets them - downset 0, upset 1)
 * ---------------------------------------------------------------------------------------- */
#define SEG_LOCUT 2.2
#define SEG_HICUT 2.5
#define SEG_MAXTRIM 100
#define SEG_DOWNSET 0
#define SEG_UPSET 1
#define LNFAC_N 2048
static double g_lnfac[LNFAC_N];       /* lnfac@0x688c40: ln(n!) printed with 6 decimals */
static double g_entray[2][13];        /* Seg::entropy_init@0x439090 for W=12 / W=8 */
static int g_seg_ready = 0;

static int seg_aaindex(unsigned char c)
{ /* Seg::getwin_init@0x4391a0: "ACDEFGHIKLMNPQRSTVWY" upper+lower -> 0..19, anything else -> -1 */
    static const char *aa = "ACDEFGHIKLMNPQRSTVWY";
    const char *p;
    if (c >= 'a' && c <= 'z') c = (unsigned char)(c - 32);
    p = strchr(aa, c);
    return (p && c) ? (int)(p - aa) : -1;
}

static void seg_init(void)
{
    int i, w;
    char buf[64];
    if (g_seg_ready) return;
    for (i = 0; i < LNFAC_N; i++) { snprintf(buf, sizeof buf, "%.6f", lgamma((double)i + 1.0)); g_lnfac[i] = atof(buf); }
    for (w = 0; w < 2; w++) {
        int W = w ? 8 : 12;
        g_entray[w][0] = 0.0;
        for (i = 1; i <= W; i++) { double p = (double)i / (double)W; g_entray[w][i] = (-p) * log(p) / LN2; }
    }
    g_seg_ready = 1;
}

typedef struct { int W, wi; } seg_par;

/* state vector = composition counts sorted descending, 0-terminated (Seg::stateon@0x4399b0) */
static void seg_state(const int *comp, int *sv)
{
    int i, j, n = 0;
    for (i = 0; i < 20; i++) if (comp[i] > 0) {
        int v = comp[i];
        for (j = n; j > 0 && sv[j - 1] < v; j--) sv[j] = sv[j - 1];
        sv[j] = v; n++;
    }
    sv[n] = 0;
}
/* Seg::entropy_cal@0x438f70 */
static double seg_entropy(const seg_par *sp, const int *sv)
{
    int total = 0, i;
    double ent = 0.0, inv;
    for (i = 0; sv[i]; i++) total += sv[i];
    if (total == sp->W) { for (i = 0; sv[i]; i++) ent += g_entray[sp->wi][sv[i]]; return ent; }
    if (total == 0) return 0.0;
    inv = 1.0 / (double)total;
    for (i = 0; sv[i]; i++) { double x = (double)sv[i]; ent += log(inv * x) * x; }
    return inv * (-ent) / LN2;
}
static void seg_comp(const char *s, int n, int *comp)
{
    int i;
    memset(comp, 0, 20 * sizeof(int));
    for (i = 0; i < n; i++) { int k = seg_aaindex((unsigned char)s[i]); if (k >= 0) comp[k]++; }
}
/* Seg::seqent@0x43a2e0: H[i] = entropy of the window that starts at i (downset 0); once the
 * window cannot shift any more the last value is repeated; -1 when the sequence is shorter. */
static double *seg_seqent(const seg_par *sp, const char *s, int n)
{
    double *H, ent;
    int comp[20], sv[21], i, start = 0, first = SEG_DOWNSET, last = n - SEG_UPSET;
    if (sp->W > n) return NULL;
    H = (double *)malloc(sizeof(double) * (size_t)n);
    for (i = 0; i < n; i++) H[i] = -1.0;
    seg_comp(s, sp->W, comp);
    seg_state(comp, sv);
    ent = seg_entropy(sp, sv);
    for (i = first; i <= last; i++) {
        H[i] = ent;
        if (start + 1 + sp->W <= n) { /* Seg::shiftwin1@0x43af30 */
            int k = seg_aaindex((unsigned char)s[start]);
            if (k >= 0) comp[k]--;
            k = seg_aaindex((unsigned char)s[start + sp->W]);
            if (k >= 0) comp[k]++;
            start++;
            seg_state(comp, sv);
            ent = seg_entropy(sp, sv);
        }
    }
    return H;
}
/* Seg::getprob@0x4393d0 = lnperm@0x438c60 + lnass@0x438c90 - total*ln(20) */
static double seg_getprob(const int *sv, int total)
{
    double ans1, ans2;
    int i;
    ans1 = g_lnfac[20];
    if (sv[0] != 0) {
        int tot = 20, cls = 1, svim1 = sv[0], svi;
        for (i = 0;; svim1 = svi) {
            if (++i == 20) { ans1 -= g_lnfac[cls]; break; }
            else if ((svi = sv[i]) == svim1) cls++;
            else {
                tot -= cls;
                ans1 -= g_lnfac[cls];
                if (svi == 0) { ans1 -= g_lnfac[tot]; break; }
                cls = 1;
            }
        }
    }
    ans2 = g_lnfac[total];
    for (i = 0; sv[i] != 0; i++) ans2 -= g_lnfac[sv[i]];
    return (ans2 + ans1) - (double)total * 2.995732273553991; /* ln 20 @0x4603d0 */
}
/* Seg::trim@0x439e20 */
static void seg_trim(const char *s, int n, int *leftend, int *rightend)
{
    int lend = 0, rend = n - 1, minlen = 1, len, i, comp[20], sv[21];
    double minprob = 1.0, prob;
    if (n - SEG_MAXTRIM > minlen) minlen = n - SEG_MAXTRIM;
    for (len = n; len > minlen; len--) {
        seg_comp(s, len, comp);
        for (i = 0;; i++) {
            seg_state(comp, sv);
            prob = seg_getprob(sv, len);
            if (prob < minprob) { minprob = prob; lend = i; rend = len + i - 1; }
            if (i + 1 + len > n) break;
            { int k = seg_aaindex((unsigned char)s[i]); if (k >= 0) comp[k]--; }
            { int k = seg_aaindex((unsigned char)s[i + len]); if (k >= 0) comp[k]++; }
        }
    }
    *leftend = *leftend + lend;
    *rightend = *rightend - (n - rend - 1);
}
/* Seg::segseq@0x43a9e0: marks mask[offset+begin .. offset+end] = 1 for every segment found */
static void seg_segseq(const seg_par *sp, const char *s, int n, int offset, uint8_t *mask)
{
    double *H = seg_seqent(sp, s, n);
    int first = SEG_DOWNSET, last = n - SEG_UPSET, lowlim = first, i, j;
    if (!H) return;
    for (i = first; i <= last; i++) {
        if (H[i] <= SEG_LOCUT && H[i] != -1.0) {
            int loi, hii, leftend, rightend;
            for (j = i; j >= lowlim; j--) { if (H[j] == -1.0) break; if (H[j] > SEG_HICUT) break; } /* findlo@0x438ba0 */
            loi = j + 1;
            for (j = i; j <= last; j++) { if (H[j] == -1.0) break; if (H[j] > SEG_HICUT) break; }   /* findhi@0x438c00 */
            hii = j - 1;
            leftend = loi - SEG_DOWNSET;
            rightend = hii + SEG_UPSET - 1;
            seg_trim(s + leftend, rightend - leftend + 1, &leftend, &rightend);
            if (i + SEG_UPSET - 1 < leftend) {
                int lend = loi - SEG_DOWNSET, rend = leftend - 1;
                seg_segseq(sp, s + lend, rend - lend + 1, offset + lend, mask);
            }
            for (j = leftend; j <= rightend; j++) mask[offset + j] = 1;
            i = (hii < rightend + SEG_DOWNSET) ? hii : rightend + SEG_DOWNSET;
            lowlim = i + 1;
        }
    }
    free(H);
}

/* ------------------------------------------------------------------------------------------
 * 6-frame translation  (BuildQHash 0x40d079-0x40d433)
 * ---------------------------------------------------------------------------------------- */
static int nt_idx(char c) { return c == 'T' ? 0 : c == 'C' ? 1 : c == 'A' ? 2 : c == 'G' ? 3 : -1; }
static char nt_comp(char c)
{ /* std::map<char,char> built in Process: ACGTU acgtu; anything else becomes 'N' (0x40d48b) */
    switch (c) {
    case 'A': return 'T'; case 'T': return 'A'; case 'C': return 'G'; case 'G': return 'C'; case 'U': return 'A';
    case 'a': return 't'; case 't': return 'a'; case 'c': return 'g'; case 'g': return 'c'; case 'u': return 'a';
    default: return 'N';
    }
}

void rs_translate6(const char *seq, int len, uint8_t *out[6], int lens[6])
{
    static const char *aa = "FFLLSSSSYY..CC.WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"; /* aa@0x688c00 */
    char *nt = (char *)malloc((size_t)len + 1), *prot = (char *)malloc((size_t)len / 3 + 2);
    uint8_t *mask = (uint8_t *)malloc((size_t)len / 3 + 2);
    int f, i;
    init_tables();
    seg_init();
    memcpy(nt, seq, (size_t)len);
    for (f = 0; f < 6; f++) {
        int o = f % 3, n = (len - o) / 3;
        seg_par sp;
        if (f == 3) { /* reverse, then complement (0x40d376-0x40d433) */
            for (i = 0; i < len / 2; i++) { char t = nt[i]; nt[i] = nt[len - 1 - i]; nt[len - 1 - i] = t; }
            for (i = 0; i < len; i++) nt[i] = nt_comp(nt[i]);
        }
        if (n < 0) n = 0;
        for (i = 0; i < n; i++) {
            int a = nt_idx(nt[o + 3 * i]), b = nt_idx(nt[o + 3 * i + 1]), c = nt_idx(nt[o + 3 * i + 2]);
            prot[i] = (a < 0 || b < 0 || c < 0) ? 'Z' : aa[16 * a + 4 * b + c]; /* UNKNOWN_AA 'Z' @0x45a1d1 */
        }
        prot[n] = 0;
        /* frames of <= 11 residues use the W=8 masker, longer ones W=12 (0x40d263-0x40d275, 0x40d4a7) */
        sp.W = (n <= 11) ? 8 : 12; sp.wi = (n <= 11) ? 1 : 0;
        memset(mask, 0, (size_t)n + 1);
        seg_segseq(&sp, prot, n, 0, mask);
        for (i = 0; i < n; i++) out[f][i] = mask[i] ? INVALID_CODE : g_code[(unsigned char)prot[i]];
        lens[f] = n;
    }
    free(nt); free(prot); free(mask);
}

/* ------------------------------------------------------------------------------------------
 * HSP records  (STResult, 0x60 bytes; fields as CalRes@0x4077a0 fills them)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int sidx, score;
    double bits, loge, ident;
    int alnlen, mism, gaps, frame, qaas, qaae, qnts, qnte, ds, de;
} hsp_t;

typedef struct { hsp_t *v; int n, cap; } hsp_vec;
static void hv_reserve(hsp_vec *h, int n) { if (n > h->cap) { h->cap = n * 2 + 16; h->v = (hsp_t *)realloc(h->v, sizeof(hsp_t) * (size_t)h->cap); } }

/* ------------------------------------------------------------------------------------------
 * std::sort / std::stable_sort of libstdc++ (GCC 4.x, pre-4.5: pivot by value, threshold 16)
 *   __introsort_loop@0x42a570 (CompObj), @0x42a0b0 (CompFrameObj), @0x42a310 (CompQSt)
 *   __final_insertion_sort@0x4263f0 ...
 * ---------------------------------------------------------------------------------------- */
typedef int (*hsp_less)(const hsp_t *, const hsp_t *);
static int less_loge(const hsp_t *a, const hsp_t *b) { return a->loge < b->loge; }   /* CompObj / CompEvalueObj: +0x10 */
static int less_frame(const hsp_t *a, const hsp_t *b) { return a->frame < b->frame; } /* CompFrameObj: +0x2c */
static int less_qst(const hsp_t *a, const hsp_t *b) { return a->qaas < b->qaas; }     /* CompQSt: +0x30 */

static void gx_push_heap(hsp_t *first, long hole, long top, hsp_t value, hsp_less lt)
{
    long parent = (hole - 1) / 2;
    while (hole > top && lt(&first[parent], &value)) { first[hole] = first[parent]; hole = parent; parent = (hole - 1) / 2; }
    first[hole] = value;
}
static void gx_adjust_heap(hsp_t *first, long hole, long len, hsp_t value, hsp_less lt)
{
    long top = hole, second = 2 * hole + 2;
    while (second < len) {
        if (lt(&first[second], &first[second - 1])) second--;
        first[hole] = first[second]; hole = second; second = 2 * (second + 1);
    }
    if (second == len) { first[hole] = first[second - 1]; hole = second - 1; }
    gx_push_heap(first, hole, top, value, lt);
}
static void gx_heapsort(hsp_t *first, hsp_t *last, hsp_less lt)
{ /* std::partial_sort(first,last,last) */
    long len = last - first, parent;
    if (len >= 2) for (parent = (len - 2) / 2;; parent--) { gx_adjust_heap(first, parent, len, first[parent], lt); if (parent == 0) break; }
    while (last - first > 1) { hsp_t v; --last; v = *last; *last = *first; gx_adjust_heap(first, 0, last - first, v, lt); }
}
static const hsp_t *gx_median(const hsp_t *a, const hsp_t *b, const hsp_t *c, hsp_less lt)
{
    if (lt(a, b)) { if (lt(b, c)) return b; else if (lt(a, c)) return c; else return a; }
    else if (lt(a, c)) return a;
    else if (lt(b, c)) return c;
    else return b;
}
static void gx_introsort_loop(hsp_t *first, hsp_t *last, long depth, hsp_less lt)
{
    while (last - first > 16) {
        hsp_t pivot, *lo, *hi;
        if (depth == 0) { gx_heapsort(first, last, lt); return; }
        --depth;
        pivot = *gx_median(first, first + (last - first) / 2, last - 1, lt);
        lo = first; hi = last;
        for (;;) { /* __unguarded_partition */
            while (lt(lo, &pivot)) ++lo;
            --hi;
            while (lt(&pivot, hi)) --hi;
            if (!(lo < hi)) break;
            { hsp_t t = *lo; *lo = *hi; *hi = t; }
            ++lo;
        }
        gx_introsort_loop(lo, last, depth, lt);
        last = lo;
    }
}
static void gx_unguarded_linear_insert(hsp_t *last, hsp_t val, hsp_less lt)
{
    hsp_t *next = last - 1;
    while (lt(&val, next)) { *last = *next; last = next; --next; }
    *last = val;
}
static void gx_insertion_sort(hsp_t *first, hsp_t *last, hsp_less lt)
{
    hsp_t *i;
    if (first == last) return;
    for (i = first + 1; i != last; ++i) {
        hsp_t val = *i;
        if (lt(&val, first)) { memmove(first + 1, first, sizeof(hsp_t) * (size_t)(i - first)); *first = val; }
        else gx_unguarded_linear_insert(i, val, lt);
    }
}
static void gx_sort(hsp_t *first, hsp_t *last, hsp_less lt)
{
    long n = last - first, lg = 0, t;
    hsp_t *i;
    if (first == last) return;
    for (t = n; t > 1; t >>= 1) lg++;
    gx_introsort_loop(first, last, 2 * lg, lt);
    if (n > 16) { gx_insertion_sort(first, first + 16, lt); for (i = first + 16; i != last; ++i) gx_unguarded_linear_insert(i, *i, lt); }
    else gx_insertion_sort(first, last, lt);
}
static void gx_stable_sort(hsp_t *first, hsp_t *last, hsp_less lt)
{ /* any stable sort gives the same permutation */
    hsp_t *i;
    for (i = first + 1; i < last; ++i) {
        hsp_t val = *i, *j = i;
        while (j > first && lt(&val, j - 1)) { *j = *(j - 1); --j; }
        *j = val;
    }
}

/* ------------------------------------------------------------------------------------------
 * Gapped X-drop extension  (CHashSearch::AlignGapped@0x40a550)
 * seq1 = query flank (n1), seq2 = subject flank (n2).  Returns the score gain; *c1,*c2 = residues
 * consumed on seq1/seq2, *ident = identities, modes/lens = run-length trace from END to START.
 * ---------------------------------------------------------------------------------------- */
typedef struct { char *m; int *l; int n, cap; } trace_t;
static void tr_push(trace_t *t, char mode)
{ /* 0x40ab0a-0x40ab98: a run continues when toupper(mode) equals toupper(last run's mode) */
    if (t->n > 0) {
        char a = mode, b = t->m[t->n - 1];
        if (a >= 'a' && a <= 'z') a = (char)(a - 32);
        if (b >= 'a' && b <= 'z') b = (char)(b - 32);
        if (a == b) { t->l[t->n - 1]++; return; }
    }
    if (t->n == t->cap) { t->cap = t->cap * 2 + 16; t->m = (char *)realloc(t->m, (size_t)t->cap); t->l = (int *)realloc(t->l, sizeof(int) * (size_t)t->cap); }
    t->m[t->n] = mode; t->l[t->n] = 1; t->n++;
}

static int align_gapped(const uint8_t *seq1, const uint8_t *seq2, int n1, int n2,
                        int *c1, int *c2, int *ident, trace_t *tr)
{
    const int open = GAP_OPEN, ext = GAP_EXT, first = GAP_OPEN + GAP_EXT;
    int *H = (int *)malloc(sizeof(int) * (size_t)(n2 + 2)), *D = (int *)malloc(sizeof(int) * (size_t)(n2 + 2));
    size_t W = (size_t)n2 + 2;
    char *pm = (char *)calloc((size_t)(n1 + 2) * W, 1), *pe = (char *)calloc((size_t)(n1 + 2) * W, 1), *pd = (char *)calloc((size_t)(n1 + 2) * W, 1);
    int jEnd = (int)((T_XGAP - (double)open) / (double)ext); /* 0x40a693-0x40a6b9 */
    int best = 0, bestI = 0, bestJ = 0, jStart = 1, i, j, r;
#define PM(i, j) pm[(size_t)(i) * W + (size_t)(j)]
#define PE(i, j) pe[(size_t)(i) * W + (size_t)(j)]
#define PD(i, j) pd[(size_t)(i) * W + (size_t)(j)]
    *c1 = *c2 = *ident = 0; tr->n = 0;
    H[0] = 0; D[0] = -open;
    PM(0, 0) = '0';
    if (n2 > 0 && jEnd > 0) { /* 0x40a6c1-0x40a770 */
        r = -open;
        for (j = 1;; ) {
            r -= ext; H[j] = r; D[j] = r - open;
            if (j == 1) { PM(0, j) = 'E'; PE(0, j) = 'E'; } else { PM(0, j) = 'e'; PE(0, j) = 'e'; }
            PD(0, j) = 'D';
            j++;
            if (jEnd < j) break;
            if (n2 < j) break;
        }
    }
    if (n1 <= 0 || jEnd <= 1) goto done; /* 0x40a782-0x40a79b -> 0x40b41a */
    for (i = 1;;) {
        int diag = H[jStart - 1], hprev, E, h = 0, Dn, grow = 1, trim = 1;
        char te, td;
        if (i == 1) { PM(i, jStart - 1) = 'D'; PD(i, jStart - 1) = 'D'; PE(i, jStart - 1) = 'E'; } /* 0x40ad90 */
        else { PM(i, jStart - 1) = 'd'; PD(i, jStart - 1) = 'd'; PE(i, jStart - 1) = 'e'; }        /* 0x40a807 */
        hprev = H[jStart - 1] - first;                                                             /* 0x40a83b-0x40a861 */
        if (hprev < D[jStart - 1] - ext) hprev = D[jStart - 1] - ext;
        D[jStart - 1] = hprev; H[jStart - 1] = hprev;
        E = hprev - open;
        if (!(jStart > jEnd) && !(n2 < jStart)) {
            for (j = jStart;;) { /* 0x40a8e4-0x40a9f5 */
                int a = hprev - first, b = E - ext, s;
                if (a >= b) { E = a; te = 'E'; } else { E = b; te = 'e'; }
                a = H[j] - first; b = D[j] - ext;
                if (a >= b) { Dn = a; td = 'D'; } else { Dn = b; td = 'd'; }
                s = diag + SUB(seq1[i - 1], seq2[j - 1]);
                h = s; PM(i, j) = 's';
                if (E > h) { PM(i, j) = te; h = E; }
                if (h < Dn) { PM(i, j) = td; h = Dn; }
                PE(i, j) = te; PD(i, j) = td;
                diag = H[j]; H[j] = h; D[j] = Dn;
                hprev = h;
                if (h > best) { best = h; bestI = i; bestJ = j; }
                else if ((double)best - T_XGAP > (double)h && j > bestJ) {
                    if (j >= jEnd) { jEnd = j; }              /* 0x40b48f: fall into the right growth */
                    else { jEnd = j; grow = 0; trim = 0; }    /* 0x40a9f5: straight to the next row */
                    break;
                }
                j++;
                if (n2 < j) break;
                if (j > jEnd) break;
            }
        }
        if (grow) { /* 0x40ac45-0x40ad08 */
            for (j = jEnd + 1; !(n2 < j); j++) {
                int a = hprev - first, b = E - ext;
                if (a > b) { E = a; te = 'E'; } else { E = b; te = 'e'; }
                PM(i, j) = te; PE(i, j) = te;
                H[j] = E; D[j] = E - open;
                if (E > best) { best = E; bestI = i; bestJ = j; }
                else if ((double)best - T_XGAP > (double)E) { jEnd = j; break; }
                hprev = E;
            }
        }
        if (trim && !(jStart > bestJ)) { /* 0x40ad0c-0x40ad82 */
            double lim = (double)best - T_XGAP;
            if (lim > (double)H[bestJ]) jStart = bestJ; /* 0x40b40b */
            else {
                int k = bestJ;
                for (;;) { k--; if (jStart > k) break; if (lim > (double)H[k]) { jStart = k; break; } }
            }
        }
        i++; /* 0x40a9fa-0x40aa1b */
        if (n1 < i) break;
        if (!(jStart < jEnd)) break;
    }
    *c1 = bestI; *c2 = bestJ;
    if (best > 0) { /* traceback 0x40aa84-0x40ac12 */
        char mode = PM(bestI, bestJ);
        i = bestI; j = bestJ;
        if (mode != 's') { fprintf(stderr, "align_gapped: trace does not start with s\n"); abort(); }
        while (!(j == 0 && i == 0)) {
            tr_push(tr, mode);
            if (mode == 's') { if (seq1[i - 1] == seq2[j - 1]) (*ident)++; i--; j--; mode = PM(i, j); }
            else if (mode == 'D' || mode == 'd') { i--; mode = (mode == 'D') ? PM(i, j) : PD(i, j); }
            else { j--; mode = (mode == 'E') ? PM(i, j) : PE(i, j); }
            if (i < 0 || j < 0) { fprintf(stderr, "align_gapped: trace ran off the matrix\n"); abort(); }
            if (mode == '0') break;
        }
    }
done:
    free(H); free(D); free(pm); free(pe); free(pd);
    return best;
#undef PM
#undef PE
#undef PD
}

/* ------------------------------------------------------------------------------------------
 * Per-query search state
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    const rs_db *db;
    int qidx, ntlen;
    ka_eff eff;
    hsp_vec res; /* the multimap<pair<q,s>,STResult>: ascending subject, newest first inside a subject */
    trace_t tr_r, tr_l;
    uint8_t *rev1, *rev2;
    int revcap;
} qstate;

/* CHashSearch::CalRes@0x4077a0 */
static void cal_res(qstate *qs, int frame, int qbeg, int sidx, int dbeg, int seedlen,
                    int score, int nmatch, int qfwd, int dfwd, int qbwd, int dbwd,
                    int alnlen, int gapopens, int gaptotal)
{
    double le = ka_loge(&qs->eff, (double)score), bits;
    hsp_t h, *v;
    int lo, hi, pos, L = qs->ntlen;
    if (le > 0.0) le = floor(le * 100.0 + 0.5) / 100.0; /* 0x4077ec-0x407829 */
    else le = floor(le * 100.0 - 0.5) / 100.0;          /* 0x407b60-0x407b8f */
    bits = floor(ka_bits((double)score) * 100.0 + 0.5) / 100.0;
    if (!(score > 30) && !(LOGE_THR > le)) return;      /* 0x4078e4-0x4078f9 */
    h.sidx = sidx; h.score = score; h.bits = bits; h.loge = le;
    h.alnlen = alnlen; h.gaps = gapopens; h.mism = alnlen - nmatch - gaptotal;
    h.ident = (double)nmatch * 100.0 / (double)alnlen;
    h.frame = frame; { static int seqno = 0; if (getenv("RS_SEQNO")) h.mism = seqno++; }
    h.qaas = qbeg - qbwd;
    h.qaae = qbeg + qfwd - 1 + seedlen;
    h.ds = dbeg - dbwd;
    h.de = dbeg + dfwd - 1 + seedlen;
    if (frame <= 2) { /* 0x407ba0-0x407bdf */
        h.qnts = (qbeg - qbwd) * 3 + frame + 1;
        h.qnte = (qbeg + qfwd + seedlen) * 3 + frame;
    } else {          /* 0x408376-0x4083ab */
        h.qnts = L - (qbeg - qbwd) * 3 - (frame - 3);
        h.qnte = h.qnts + 1 - (qfwd + qbwd + seedlen) * 3;
    }
    /* lower_bound on subject (0x4082b0-0x40831e), duplicate test against that one element
     * (0x4083b0-0x408446), otherwise insert in front of it (_M_insert_equal_ with hint) */
    v = qs->res.v; lo = 0; hi = qs->res.n;
    while (lo < hi) { int mid = (lo + hi) / 2; if (v[mid].sidx < sidx) lo = mid + 1; else hi = mid; }
    pos = lo;
    if (pos < qs->res.n && v[pos].sidx == sidx && v[pos].frame == frame && v[pos].qaas == h.qaas &&
        v[pos].ds == h.ds && v[pos].qaae == h.qaae && v[pos].de == h.de) {
        if (v[pos].loge > h.loge) {
            v[pos].score = h.score; v[pos].bits = h.bits; v[pos].loge = h.loge; v[pos].ident = h.ident;
            v[pos].alnlen = h.alnlen; v[pos].mism = h.mism; v[pos].gaps = h.gaps; v[pos].qnts = h.qnts; v[pos].qnte = h.qnte;
        }
        return;
    }
    hv_reserve(&qs->res, qs->res.n + 1);
    v = qs->res.v;
    memmove(v + pos + 1, v + pos, sizeof(hsp_t) * (size_t)(qs->res.n - pos));
    v[pos] = h; qs->res.n++;
}

/* CHashSearch::AlignSeqs@0x413370 followed by the CalRes call of ExtendSeq2Set 0x414150-0x41422a.
 * q/qlen: frame codes; qpos: start of the (grown) seed; d/dlen/dpos likewise; L grown seed length. */
static void align_and_record(qstate *qs, int frame, const uint8_t *q, int qlen, int qpos,
                             int sidx, const uint8_t *d, int dlen, int dpos, int L, int score, int nmatch)
{
    int qfwd = 0, dfwd = 0, qbwd = 0, dbwd = 0, s0 = score, total, i;
    int ungapped_len, alnlen, gapopens = 0, gaptotal = 0;
    int nr = 0, nl = 0; /* gapped runs on the right / left */
    /* forward ungapped x-drop (0x4136e0-0x4137a7) */
    {
        int n1 = qlen - qpos - L, n2 = dlen - dpos - L, gain = 0, bl = 0, bi = 0;
        if (n1 != 0 && n2 != 0 && !(s0 < -20)) {
            const uint8_t *p1 = q + qpos + L, *p2 = d + dpos + L;
            int run = s0, best = s0, id = 0;
            for (i = 0;;) {
                run += SUB(p1[i], p2[i]); id += (p1[i] == p2[i]); i++;
                if (run > best) { best = run; bl = i; bi = id; }
                if (!(n2 > i)) break;
                if (n1 <= i) break;
                if (run < -20) break;
                if ((double)run < (double)best - T_XUNGAP) break;
            }
            gain = best - s0;
        }
        nmatch += bi; qfwd += bl; dfwd += bl; score = s0 + gain;
        total = gain;
    }
    /* backward ungapped x-drop from the seed score (0x413600-0x4136d7) */
    {
        int a = qpos - 1, b = dpos - 1, bl = 0, bi = 0, gain = 0;
        if (a >= 0 && b >= 0 && !(s0 < -20)) {
            int run = s0, best = s0, id = 0, cnt = 0;
            for (;;) {
                run += SUB(q[a], d[b]); id += (q[a] == d[b]); cnt++;
                if (best < run) { best = run; bl = cnt; bi = id; }
                a--; b--;
                if (b < 0) break;
                if (a < 0) break;
                if (run < -20) break;
                if ((double)run < (double)best - T_XUNGAP) break;
            }
            gain = best - s0;
        }
        nmatch += bi; qbwd += bl; dbwd += bl; score = s0 + gain + total;
    }
    ungapped_len = qfwd + L + qbwd;
    /* gapped extension of both flanks (0x4134c8-0x413b38) */
    if (!(T_GAPTRIG > (double)score)) {
        int qend = qfwd + qpos + L, dend = dfwd + dpos + L;
        int dright = dlen - dend, qright = qlen - qend, c1, c2, id, g;
        if (dright > 2 && qright > 2) {
            g = align_gapped(q + qend, d + dend, qright, dright, &c1, &c2, &id, &qs->tr_r);
            if (g > 0) { score += g; nmatch += id; qfwd += c1; dfwd += c2; nr = qs->tr_r.n; }
        }
        {
            int dleft = dpos - dbwd, qleft = qpos - qbwd;
            if (dleft > 2 && qleft > 2) {
                int need = (qleft > dleft ? qleft : dleft) + 1;
                if (need > qs->revcap) { qs->revcap = need * 2; qs->rev1 = (uint8_t *)realloc(qs->rev1, (size_t)qs->revcap); qs->rev2 = (uint8_t *)realloc(qs->rev2, (size_t)qs->revcap); }
                for (i = 0; i < qleft; i++) qs->rev1[i] = q[qleft - 1 - i];
                for (i = 0; i < dleft; i++) qs->rev2[i] = d[dleft - 1 - i];
                g = align_gapped(qs->rev1, qs->rev2, qleft, dleft, &c1, &c2, &id, &qs->tr_l);
                if (g > 0) { score += g; nmatch += id; qbwd += c1; dbwd += c2; nl = qs->tr_l.n; }
            }
        }
    }
    /* CalRes 0x40786f-0x4078d1: alnlen = sum of run lengths, every non-'s' run is one gap opening */
    alnlen = ungapped_len;
    for (i = 0; i < nr; i++) { alnlen += qs->tr_r.l[i]; if (qs->tr_r.m[i] != 's') { gapopens++; gaptotal += qs->tr_r.l[i]; } }
    for (i = 0; i < nl; i++) { alnlen += qs->tr_l.l[i]; if (qs->tr_l.m[i] != 's') { gapopens++; gaptotal += qs->tr_l.l[i]; } }
    cal_res(qs, frame, qpos, sidx, dpos, L, score, nmatch, qfwd, dfwd, qbwd, dbwd, alnlen, gapopens, gaptotal);
}

/* suffix-key helpers (ExtendSeq2Set 0x413bd2-0x414aa1) */
static int klen(unsigned k)
{
    int low = (k & 0xf) != 0xf;
    int l = ((k & 0xff) != 0xff) ? 3 + low : 2 + low;
    l -= ((k & 0xfff) == 0xfff);
    l -= (k == 0xffff);
    return l;
}
static int key_lb_less(unsigned dbk, unsigned qk)
{ /* std::lower_bound comparator, 0x413ce0-0x413d8e */
    int ld = klen(dbk), lq = klen(qk), n = ld < lq ? ld : lq;
    if (n != 0) { int sh = (4 - n) * 4; int a = (int)dbk >> sh, b = (int)qk >> sh; if (a != b) return a < b; }
    return ld < lq;
}
static int key_ub_less(unsigned qk, unsigned dbk)
{ /* std::upper_bound comparator, 0x414785-0x41483e */
    int ld = klen(dbk), lq = klen(qk), n = lq <= ld ? lq : ld;
    if (n == 0) return lq < ld;
    { int sh = (4 - n) * 4; int a = (int)qk >> sh, b = (int)dbk >> sh; if (a == b) return 0; return a < b; }
}

/* CHashSearch::ExtendSeq2Set@0x413b90 */
static int extend_seq2set(qstate *qs, int seed, int seedlen, const uint8_t *key, int nkey,
                          int frame, const uint8_t *q, int qlen, int qpos)
{
    const rs_db *db = qs->db;
    int64_t b0 = db->bstart[seed];
    int n = (int)(db->bstart[seed + 1] - b0), nst = 0, ned = n, i;
    const uint32_t *post = db->post + b0;
    const uint16_t *keys = db->keys + b0;
    if (seedlen > 6) {
        unsigned qk = 0;
        int lo, len, m;
        if (nkey == 0) return 0;
        for (i = 0; i < nkey; i++) qk |= (unsigned)key[i] << (12 - 4 * i);
        for (i = nkey; i <= 3; i++) qk |= 0xfu << (12 - 4 * i);
        qk &= 0xffff;
        lo = 0; len = n;
        while (len > 0) { int half = len >> 1; if (key_lb_less(keys[lo + half], qk)) { lo += half + 1; len -= half + 1; } else len = half; }
        nst = lo;
        if (nst == n) return 0;
        m = klen(qk) < klen(keys[nst]) ? klen(qk) : klen(keys[nst]);
        if (m == 0) return 0;
        { int sh = (4 - m) * 4; if (((int)keys[nst] >> sh) != ((int)qk >> sh)) return 0; }
        lo = 0; len = n;
        while (len > 0) { int half = len >> 1; if (key_ub_less(qk, keys[lo + half])) len = half; else { lo += half + 1; len -= half + 1; } }
        ned = lo;
    }
    for (i = nst; i < ned; i++) {
        unsigned p = post[i];
        int dpos = (int)(p & 0x7ff), sidx = (int)(p >> 11);
        int dlen = (int)(db->off[sidx + 1] - db->off[sidx]);
        const uint8_t *d = db->res + db->off[sidx];
        int score = 0, ident = 0, L, k, lim, qp = qpos, dp = dpos, back;
        if ((unsigned)(dpos + seedlen) > (unsigned)dlen) continue;                       /* 0x413e96-0x413e9e */
        if (qpos != 0 && dpos != 0 && g_grp[q[qpos - 1]] == g_grp[d[dpos - 1]] && nkey != 4) continue; /* 0x4140c0-0x414132 */
        for (k = 0; k < seedlen; k++) { score += SUB(q[qpos + k], d[dpos + k]); ident += (q[qpos + k] == d[dpos + k]); }
        L = seedlen;
        lim = dlen - dpos; if (lim > qlen - qpos) lim = qlen - qpos;                    /* 0x413fd2-0x413fe3 */
        while (lim > L && g_grp[q[qpos + L]] == g_grp[d[dpos + L]]) {                   /* 0x413fe8-0x4142e3 */
            score += SUB(q[qpos + L], d[dpos + L]); ident += (q[qpos + L] == d[dpos + L]); L++;
        }
        back = qpos < dpos ? qpos : dpos;                                               /* 0x414020-0x41438f */
        while (back > 0 && g_grp[q[qp - 1]] == g_grp[d[dp - 1]]) {
            qp--; dp--; back--; L++;
            score += SUB(q[qp], d[dp]); ident += (q[qp] == d[dp]);
        }
        if ((double)score >= T_SEEDSCORE && ident >= T_SEEDIDENT)                       /* 0x414058-0x414073 */
            align_and_record(qs, frame, q, qlen, qp, sidx, d, dlen, dp, L, score, ident);
    }
    return ned - nst;
}

/* CHashSearch::Searching@0x415050, one frame */
static void search_frame(qstate *qs, int frame, const uint8_t *q, int qlen)
{
    const rs_db *db = qs->db;
    static const int strides[3] = {10, 1, 100}; /* this+0x40400, pushed in Process 0x41b609-0x41b748 */
    int pos, prev = 6;
    uint8_t key[8];
    if (qlen <= 6) return; /* 0x4151fa-0x415201, 0x4152e7: frames of < 6 and of exactly 6 residues are skipped */
    for (pos = 0; pos + 6 < qlen; pos++) { /* 0x415d60-0x415d73: the last window is never visited */
        int seed = 0, k, len = 6, used, bad = 0, r, g;
        unsigned freq;
        for (k = 0; k < 6; k++) { g = g_grp[q[pos + k]]; if (g == INVALID_GRP) { bad = 1; break; } seed = seed * 10 + g; }
        if (bad) continue;
        freq = (unsigned)(db->bstart[seed + 1] - db->bstart[seed]);
        if (freq > db->freq_thr) { /* 0x415ec0-0x415f71 */
            int rest = qlen - pos - 6, maxextra = (rest >= 2) ? 3 : rest + 1;
            if (maxextra <= 1) len = 7;
            else {
                double e, thr = (double)db->freq_thr;
                int extra = 1, idx = pos + 7;
                g = g_grp[q[pos + 6]];
                if (g == INVALID_GRP) continue;
                e = (double)freq * db->letter_p[g];
                if (!(thr >= e)) {
                    for (;;) {
                        extra++;
                        if (!(maxextra > extra)) break;
                        g = g_grp[q[idx]];
                        if (g == INVALID_GRP) { bad = 1; break; }
                        idx++;
                        e *= db->letter_p[g];
                        if (thr >= e) break;
                    }
                    if (bad) continue;
                }
                len = 6 + extra;
            }
        }
        used = (len >= prev - 1) ? len : prev - 1; /* 0x4153e6-0x4153f2 */
        if (qlen < pos + used) continue;            /* 0x415404-0x41540c */
        for (k = 6; k < used; k++) key[k - 6] = g_grp[q[pos + k]];
        if (freq != 0) {
            r = extend_seq2set(qs, seed, used, key, used - 6, frame, q, qlen, pos);
            prev = used;
            if (r <= 0) prev = 6;
        }
        /* one-substitution neighbourhood, 10-mers (0x41557a-0x415d39) */
        if (!(qlen < pos + 10)) {
            int ok = 1, m;
            for (k = pos + used; k < pos + 10; k++) if (g_grp[q[k]] == INVALID_GRP) { ok = 0; break; }
            if (!ok) continue;
            for (k = 0; k < 4; k++) key[k] = g_grp[q[pos + 6 + k]];
            for (m = 0; m < 3; m++) {
                int st = strides[m], start = seed - ((seed / st) % 10) * st, j;
                for (j = 0; j < 10; j++) {
                    int v = start + j * st;
                    if (v == seed) continue;
                    if (db->bstart[v + 1] != db->bstart[v]) extend_seq2set(qs, v, 10, key, 4, frame, q, qlen, pos);
                }
            }
            /* then the same bucket with the first key residue substituted (0x416020-0x4165d6):
             * a 10-mer with one reduced-alphabet substitution at offset 6 */
            {
                int orig = key[0];
                for (k = 0; k < 10; k++) {
                    if (k == orig) continue;
                    key[0] = (uint8_t)k;
                    extend_seq2set(qs, seed, 10, key, 4, frame, q, qlen, pos);
                }
            }
        }
    }
}

/* CHashSearch::SumEvalue@0x408a50 on v[st,ed) */
static void sum_evalue(qstate *qs, hsp_vec *v, int st, int ed, int subj_len)
{
    hsp_t *a = v->v + st, *res, *chosen;
    int n = ed - st, part, nres = 0, pass, i, j;
    gx_sort(a, a + n, less_frame);
    for (part = 0; part < n && !(a[part].frame > 2); part++) {}
    if ((n - part) <= 1 && part <= 1) return;
    res = (hsp_t *)malloc(sizeof(hsp_t) * (size_t)n);
    chosen = (hsp_t *)malloc(sizeof(hsp_t) * (size_t)n);
    for (pass = 0; pass < 2; pass++) {
        hsp_t *g = pass ? a + part : a;
        int gn = pass ? n - part : part, nc = 0;
        if (gn == 0) continue;
        if (gn == 1) { if (LOGE_THR > g[0].loge) res[nres++] = g[0]; continue; }
        gx_sort(g, g + gn, less_qst);
        gx_stable_sort(g, g + gn, less_loge);
        chosen[nc++] = g[0];
        for (i = 1; i < gn; i++) {
            hsp_t *e = &g[i];
            int ov = (e->qaae + 1 - e->qaas) >> 1, ok = 1;
            if (ov > 10) ov = 10;
            if (e->loge >= 1.0 && !(e->score > 30)) continue; /* 0x408d9a-0x408d5c */
            for (j = 0; j < nc; j++) {
                hsp_t *c = &chosen[j];
                if (e->qaas <= c->qaae - ov) { if (e->qaae >= c->qaas + ov) { ok = 0; break; } }
                if (e->qaae - ov < c->qaas) continue;
                if (c->qaae >= ov + e->qaas) { ok = 0; break; }
            }
            if (ok) chosen[nc++] = *e;
        }
        if (nc == 1) { if (LOGE_THR > chosen[0].loge) res[nres++] = chosen[0]; }
        else {
            double sc[5], E, le;
            int k = nc < 5 ? nc : 5;
            for (i = 0; i < k; i++) sc[i] = (double)chosen[i].score;
            E = ka_sum_expect(&qs->eff, k, sc, subj_len);
            le = (E == 0.0) ? -10000.0 : log(E) / 2.302585092994046;
            if (LOGE_THR > le) for (i = 0; i < nc; i++) { chosen[i].loge = le; res[nres++] = chosen[i]; }
        }
    }
    if (nres > 0) { /* 0x408f98-0x409086: v = v[:st] + res + v[ed:] */
        memmove(v->v + st + nres, v->v + ed, sizeof(hsp_t) * (size_t)(v->n - ed));
        memcpy(v->v + st, res, sizeof(hsp_t) * (size_t)nres);
        v->n = v->n - (ed - st) + nres;
    }
    free(res); free(chosen);
}

/* CHashSearch::MergeRes@0x40e3b0 (m8 pass 0x40ed5e-0x40f007): the rows of a query are read back from
 * the per-thread temp file, keyed by the log(E) TEXT field re-parsed with lexical_cast
 * (CMergeUnit::Update@0x43b170, 0x43b970-0x43bb76), and put through
 * std::partial_sort(first, first+min(n,500), last) = make_heap + sort_heap
 * (__adjust_heap<CSortUnit>@0x4200c0).  Rows arrive already ascending, so the only visible effect is
 * the deterministic (unstable) permutation heap-sort applies to rows whose printed log(E) is equal. */
static void mr_adjust_heap(rs_row *a, double *k, long hole, long len, rs_row v, double kv)
{
    long top = hole, sc = hole, parent;
    while (sc < (len - 1) / 2) {
        sc = 2 * (sc + 1);
        if (k[sc] < k[sc - 1]) sc--;
        a[hole] = a[sc]; k[hole] = k[sc]; hole = sc;
    }
    if ((len & 1) == 0 && sc == (len - 2) / 2) {
        sc = 2 * (sc + 1);
        a[hole] = a[sc - 1]; k[hole] = k[sc - 1]; hole = sc - 1;
    }
    parent = (hole - 1) / 2;
    while (hole > top && k[parent] < kv) { a[hole] = a[parent]; k[hole] = k[parent]; hole = parent; parent = (hole - 1) / 2; }
    a[hole] = v; k[hole] = kv;
}
static void merge_res_order(rs_row *a, int n)
{
    double *k;
    long i, m;
    char buf[64];
    if (n < 2) return;
    k = (double *)malloc(sizeof(double) * (size_t)n);
    for (i = 0; i < n; i++) { snprintf(buf, sizeof buf, "%g", a[i].loge); k[i] = atof(buf); }
    for (i = (n - 2) / 2;; i--) { mr_adjust_heap(a, k, i, n, a[i], k[i]); if (i == 0) break; }
    for (m = n; m > 1;) { rs_row v; double kv; m--; v = a[m]; kv = k[m]; a[m] = a[0]; k[m] = k[0]; mr_adjust_heap(a, k, 0, m, v, kv); }
    free(k);
}

/* CHashSearch::PrintRes@0x409310 (with -b 0: only the m8 loop 0x409db8-0x409fd7 emits rows) */
static int print_res(qstate *qs, rs_row *rows, int max_rows)
{
    hsp_vec v = {0};
    int i, nout = 0, grp_start = 0, cur;
    if (qs->res.n == 0) return 0;
    hv_reserve(&v, qs->res.n);
    cur = qs->res.v[0].sidx;
    v.v[v.n++] = qs->res.v[0];
    for (i = 1; i < qs->res.n; i++) {
        int s = qs->res.v[i].sidx;
        if (s != cur) {
            if (v.n - grp_start > 1) sum_evalue(qs, &v, grp_start, v.n, rs_db_seqlen(qs->db, cur));
            grp_start = v.n; cur = s;
        }
        v.v[v.n++] = qs->res.v[i];
    }
    if (v.n - grp_start > 1) sum_evalue(qs, &v, grp_start, v.n, rs_db_seqlen(qs->db, cur));
    gx_sort(v.v, v.v + v.n, less_loge);
    for (i = 0; i < v.n && i < MAX_M8 && nout < max_rows; i++) {
        hsp_t *h = &v.v[i];
        rs_row *r;
        if (!(h->loge < LOGE_THR)) break;
        r = &rows[nout++];
        r->query = qs->qidx; r->subject = h->sidx; r->ident = h->ident; r->alnlen = h->alnlen; r->mismatch = h->mism;
        r->gapopen = h->gaps; r->qstart = h->qnts; r->qend = h->qnte; r->sstart = h->ds; r->send = h->de;
        r->loge = h->loge; r->bits = h->bits; r->score = h->score; r->frame = h->frame;
    }
    free(v.v);
    merge_res_order(rows, nout);
    return nout;
}

int rs_search_read(const rs_db *db, int query_index, const char *seq, int len, rs_row *rows, int max_rows)
{
    qstate qs;
    uint8_t *fr[6];
    int lens[6], f, n;
    memset(&qs, 0, sizeof qs);
    qs.db = db; qs.qidx = query_index; qs.ntlen = len;
    for (f = 0; f < 6; f++) fr[f] = (uint8_t *)malloc((size_t)len / 3 + 2);
    rs_translate6(seq, len, fr, lens);
    /* the length adjustment is taken from the first frame of each strand = len/3 (0x41526c-0x4152a4) */
    qs.eff = ka_effective(db, lens[0]);
    for (f = 0; f < 6; f++) search_frame(&qs, f, fr[f], lens[f]);
    n = print_res(&qs, rows, max_rows);
    for (f = 0; f < 6; f++) free(fr[f]);
    free(qs.res.v); free(qs.tr_r.m); free(qs.tr_r.l); free(qs.tr_l.m); free(qs.tr_l.l); free(qs.rev1); free(qs.rev2);
    return n;
}

int rs_format_row(const rs_db *db, const rs_row *r, const char *qname, char *buf, int buflen)
{
    return snprintf(buf, (size_t)buflen, "%s\t%s\t%g\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%g\t%g\n", qname, db->names[r->subject],
                    r->ident, r->alnlen, r->mismatch, r->gapopen, r->qstart, r->qend, r->sstart, r->send, r->loge, r->bits);
}

#ifdef RS_MAIN
/* rs_port <rapdb> <reads.fa> <out.m8>   - FASTA with one sequence line per record or multi-line */
int main(int argc, char **argv)
{
    rs_db *db;
    FILE *f, *o;
    char *line = NULL, **names = NULL, **seqs = NULL;
    size_t cap = 0;
    long nr = 0, capn = 0, i;
    ssize_t l;
    char **outbuf;
    if (argc < 4) { fprintf(stderr, "usage: %s rapdb reads.fa out.m8\n", argv[0]); return 2; }
    db = rs_db_load_rapdb(argv[1]);
    if (!db) { fprintf(stderr, "cannot load %s\n", argv[1]); return 1; }
    f = fopen(argv[2], "r");
    if (!f) { fprintf(stderr, "cannot open %s\n", argv[2]); return 1; }
    while ((l = getline(&line, &cap, f)) > 0) {
        while (l > 0 && (line[l - 1] == '\n' || line[l - 1] == '\r')) line[--l] = 0;
        if (line[0] == '>') {
            char *sp;
            if (nr == capn) { capn = capn * 2 + 1024; names = (char **)realloc(names, sizeof(char *) * (size_t)capn); seqs = (char **)realloc(seqs, sizeof(char *) * (size_t)capn); }
            sp = strpbrk(line, " \t"); if (sp) *sp = 0;
            names[nr] = strdup(line + 1); seqs[nr] = strdup(""); nr++;
        } else if (nr > 0) {
            size_t a = strlen(seqs[nr - 1]);
            seqs[nr - 1] = (char *)realloc(seqs[nr - 1], a + (size_t)l + 1);
            memcpy(seqs[nr - 1] + a, line, (size_t)l + 1);
        }
    }
    fclose(f);
    outbuf = (char **)calloc((size_t)nr, sizeof(char *));
#pragma omp parallel for schedule(dynamic, 64)
    for (i = 0; i < nr; i++) {
        rs_row rows[MAX_M8];
        int n = rs_search_read(db, (int)i, seqs[i], (int)strlen(seqs[i]), rows, MAX_M8), k, p = 0;
        if (n > 0) {
            outbuf[i] = (char *)malloc((size_t)n * 256);
            for (k = 0; k < n; k++) p += rs_format_row(db, &rows[k], names[i], outbuf[i] + p, 256);
        }
    }
    o = fopen(argv[3], "w");
    for (i = 0; i < nr; i++) if (outbuf[i]) { fputs(outbuf[i], o); free(outbuf[i]); }
    fclose(o);
    return 0;
}
#endif
