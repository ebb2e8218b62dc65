// k_grid.h - the training workflow's grid classification over the m8 rows (training/training.py:311-334).
#pragma once
#include "mc_hip_common.h"

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
