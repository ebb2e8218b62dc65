// tests/emul/mc_emul.cpp - TEST-ONLY sequential emulation of the GPU pipeline.
//
// Compiles the per-thread device functions of microbecensus_amd/csrc (mc_core.h, mc_finish.h) with g++ and
// runs them one "thread" at a time, stage by stage, exactly in the decomposition the HIP kernels use
// (translate+SEG -> seed enumeration -> seed evaluation -> gapped extension -> sort by (read,subject,chrono)
// -> per-read finishing).  It exists so the kernel logic can be checked against the oracle / golden m8 on
// machines without a GPU.  It is not part of the product and is never loaded by microbecensus_amd.
//
//   mc_emul <markers.faa> <reads.fa> <out.m8> [rapdb-to-verify-index]
#include "../../microbecensus_amd/csrc/mc_finish.h"
#include "../../microbecensus_amd/csrc/mc_index.h"
#include <algorithm>
#include <cstdio>
#include <fstream>
#include <iostream>

static bool read_fasta(const char *path, std::vector<std::string> &names, std::vector<std::string> &seqs)
{
    std::ifstream f(path);
    if (!f) return false;
    std::string line;
    while (std::getline(f, line)) {
        while (!line.empty() && (line.back() == '\r' || line.back() == '\n')) line.pop_back();
        if (line.empty()) continue;
        if (line[0] == '>') { size_t e = line.find_first_of(" \t"); names.push_back(line.substr(1, e == std::string::npos ? e : e - 1)); seqs.push_back(""); }
        else if (!seqs.empty()) seqs.back() += line;
    }
    return true;
}

struct Emit {
    std::vector<McSeedTask> *out; uint32_t read; int frame; const McIndex *X;
    void operator()(int bucket, int nst, int cnt, int seedlen, int nkey, int pos, int phase)
    {
        uint32_t b0 = X->bstart[bucket];
        for (int i = 0; i < cnt; i++) {
            McSeedTask t; t.read = read; t.chrono = MC_CHRONO(frame, pos, phase, nst + i); t.posting = X->post[b0 + nst + i];
            t.seedlen_nkey = (uint32_t)seedlen | ((uint32_t)nkey << 8);
            out->push_back(t);
        }
    }
};

int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: %s markers.faa reads.fa out.m8 [rapdb]\n", argv[0]); return 2; }
    std::vector<std::string> mn, ms, rn, rs;
    if (!read_fasta(argv[1], mn, ms) || !read_fasta(argv[2], rn, rs)) { fprintf(stderr, "cannot read input\n"); return 1; }
    std::vector<const char *> np, sp;
    for (size_t i = 0; i < mn.size(); i++) { np.push_back(mn[i].c_str()); sp.push_back(ms[i].c_str()); }
    McHostIndex H; std::string err;
    if (!mc_build_index(H, np.data(), sp.data(), (int)mn.size(), err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    fprintf(stderr, "index: %d seqs %ld residues %zu postings thr %u\n", H.nseq, (long)H.nres, H.post.size(), H.freq_thr);
    if (getenv("MC_DUMP_INDEX")) {   // raw arrays for tests/test_index.py: bstart, post, keys, codes, thr, letter_p
        FILE *d = fopen(getenv("MC_DUMP_INDEX"), "wb");
        uint64_t np_ = H.post.size(), nr_ = H.res_code.size();
        fwrite(&np_, 8, 1, d); fwrite(&nr_, 8, 1, d);
        fwrite(H.bstart.data(), 4, H.bstart.size(), d); fwrite(H.post.data(), 4, np_, d); fwrite(H.keys.data(), 2, np_, d); fwrite(H.res_code.data(), 1, nr_, d);
        fwrite(&H.freq_thr, 4, 1, d); fwrite(H.letter_p, 8, 10, d);
        fclose(d);
    }
    if (getenv("MC_CHECK_SCAN")) {   // exhaustive: scan-based ranges == binary-search ranges for every key a query could match
        McIndex Xc; Xc.res = H.res.data(); Xc.off = H.off.data(); Xc.bstart = H.bstart.data(); Xc.post = H.post.data(); Xc.keys = H.keys.data(); Xc.rec = H.rec.data(); Xc.filt = H.filt.data(); Xc.wild = H.wild.data(); Xc.pair = H.pair.data(); Xc.rt = H.rt.data(); Xc.rt_mask = H.rt_mask; Xc.nseq = H.nseq;
        if (H.rec.empty()) { fprintf(stderr, "scan check: the index has no bucket records\n"); return 3; }
        fprintf(stderr, "largest bucket: %u postings\n", H.max_bucket);
        long checked = 0, bad = 0, fneg = 0, fq = 0, f9q = 0, f9pos = 0, wq = 0, wpos = 0, rtq = 0, rtbad = 0, pq = 0, ppos = 0, gmq = 0, gmbad = 0;
        for (int b = 0; b < MC_NBUCKET; b++) {
            uint32_t n = H.bstart[b + 1] - H.bstart[b];
            for (uint32_t i = 0; i < n; i++) {
                uint32_t k = H.keys[H.bstart[b] + i];
                uint32_t cand[8]; int nc = 0;
                cand[nc++] = k; cand[nc++] = k | 0xF;                       // its 4- and 3-nibble query forms
                cand[nc++] = (k & 0xFFF0) | ((k + 1) & 0xF); cand[nc++] = ((k + 0x1000) & 0xF000) | (k & 0x0FFF);   // near misses
                cand[nc++] = (k & 0xFF0F) | 0xA0; cand[nc++] = (k & 0xF0FF) | 0x0A00; cand[nc++] = (k & 0x0FFF) | 0xA000;                                // keys holding the invalid group
                for (int c = 0; c < nc; c++) {
                    uint32_t qk = cand[c] & 0xFFFF;
                    if (mc_klen(qk) < 3) continue;
                    McSeedCount s1{0, 0, 0}, s2{0, 0, 0}; int n1 = 0, n2 = 0;
                    int r1 = mc_key_range(Xc, b, qk, &n1, &s1), r2 = mc_key_range_rec(Xc.rec, Xc.keys, b, qk, &n2, &s2);
                    checked++;
                    if (r1 != r2 || (r1 > 0 && n1 != n2) || s1.keyprobes != s2.keyprobes || s1.lookups != s2.lookups) bad++;
                    if (mc_klen(qk) >= 3 && (mc_klen(qk) == 4 || (qk & 0xF) == 0xF)) {   // probe forms: in a long group the range table answers
                        const McBucketRec &RR = H.rec[b];
                        const int k6 = (int)(qk >> 12);
                        if (k6 <= 10 && (int)RR.cum[k6 + 1] - (int)RR.cum[k6] <= 8 && (int)RR.cum[k6 + 1] > (int)RR.cum[k6]) {   // short group: the probe form of the scan
                            int la = 0, lb2 = 0;
                            const int ns = (int)RR.cum[k6 + 1] - (int)RR.cum[k6];
                            const int ca = mc_group_range8(H.keys.data() + RR.start + RR.cum[k6], ns, qk, &la), cb = mc_group_match8(H.keys.data() + RR.start + RR.cum[k6], ns, qk, &lb2);
                            gmq++;
                            if (ca != cb || (ca > 0 && la != lb2)) gmbad++;
                        }
                        if ((int)RR.cum[k6 + 1] - (int)RR.cum[k6] > 8) {
                            int n3 = 0;
                            const int r3 = mc_rt_lookup(H.rt.data(), H.rt_mask, (uint32_t)b, qk, &n3);
                            rtq++;
                            if (r3 != r1 || (r1 > 0 && n3 != n1)) rtbad++;
                        }
                    }
                    if (mc_klen(qk) == 3 && (qk & 0xF) == 0xF) {              // exact 9-mer probe: its filter may not lose a range either
                        const uint32_t hh = mc_filter_hash((uint32_t)b, qk), bits = mc_filter_bits(hh);
                        const bool pass = (H.filt[mc_filter9_word(hh)] & bits) == bits;
                        f9q++;
                        if (r1 > 0 && !pass) fneg++;
                        if (r1 == 0 && pass) f9pos++;
                    }
                    if (mc_klen(qk) == 4) {                                   // 10-mer probe: its filters may not lose a range
                        fq++;
                        // wildcard filter: a 10-mer with a range answers yes for all four wildcard positions
                        const uint32_t ctx = mc_wild_ctx((uint32_t)b, qk), line = mc_wild_line(ctx);
                        for (int g = 0; g < 4; g++) {
                            const bool w = mc_wild_test(&H.wild[(size_t)line * MC_WILD_LINE_WORDS + (size_t)g * 2], mc_wild_bits(ctx, (uint32_t)b, qk, g));
                            wq++;
                            if (r1 > 0 && !w) fneg++;
                            if (r1 == 0 && w) wpos++;
                            // pair filter: the residue at the wildcard offset is among the answers of its (context, offset) block
                            const uint32_t hp = mc_pair_hash((uint32_t)b, qk, g), dj = mc_pair_digit((uint32_t)b, qk, g);
                            if (dj <= 9) {
                                const uint32_t *pb = &H.pair[(size_t)mc_pair_block(hp) * 4];
                                const bool pp = (mc_pair_test4(pb[0], pb[1], pb[2], pb[3], hp) >> dj) & 1u;
                                pq++;
                                if (r1 > 0 && !pp) fneg++;
                                if (r1 == 0 && pp) ppos++;
                            }
                        }
                    }
                }
            }
        }
        fprintf(stderr, "scan check: %ld probes, %ld mismatches\n", checked, bad);
        { long set9 = 0; for (uint32_t i = 0; i < MC_FILT9_WORDS; i++) set9 += __builtin_popcount(H.filt[i]);
          fprintf(stderr, "filter check: %ld 10-mer and %ld 9-mer probes, %ld false negatives, %ld false positives of the 9-mer filter among near misses, %.1f %% of its bits set\n", fq, f9q, fneg, f9pos, 100.0 * (double)set9 / (32.0 * MC_FILT9_WORDS));
          long setw = 0; for (uint32_t w : H.wild) setw += __builtin_popcount(w);
          fprintf(stderr, "wildcard filter: %ld questions, %ld positive among near misses, %.1f %% of the bits set\n", wq, wpos, 100.0 * (double)setw / (32.0 * MC_WILD_LINE_WORDS * MC_WILD_LINES));
          long setp = 0; for (uint32_t w : H.pair) setp += __builtin_popcount(w);
          fprintf(stderr, "pair filter: %ld questions, %ld positive among near misses, %.1f %% of the cell bits set\n", pq, ppos, 100.0 * (double)setp / (120.0 * MC_PAIR_BLOCKS)); }
        { size_t used = 0; for (unsigned long long e : H.rt) used += (e != ~0ull); fprintf(stderr, "range table: %zu entries in %zu slots; %ld probes into long groups checked, %ld differ from the binary searches\n", used, H.rt.size(), rtq, rtbad); }
        fprintf(stderr, "probe form of the group scan: %ld probes, %ld differ from the general form\n", gmq, gmbad);
        return (bad || fneg || rtbad || gmbad) ? 3 : 0;
    }
    int read_len = rs.empty() ? 0 : (int)rs[0].size();
    static McTables T;
    mc_fill_tables(T, H, read_len, getenv("MC_LOGE_THR") ? atof(getenv("MC_LOGE_THR")) : 1.0);
    McIndex X; X.res = H.res.data(); X.off = H.off.data(); X.bstart = H.bstart.data(); X.post = H.post.data(); X.keys = H.keys.data(); X.rec = H.rec.empty() ? nullptr : H.rec.data(); X.filt = H.filt.data(); X.wild = H.wild.data(); X.pair = H.pair.data(); X.rt = H.rt.data(); X.rt_mask = H.rt_mask; X.nseq = H.nseq;
    McClassPars P; memset(&P, 0, sizeof P); P.nfam = 1; P.read_len = read_len;
    std::vector<int32_t> fam(H.nseq, 0);

    const int FP = MC_MAXAA + 2;
    std::vector<uint8_t> frames((size_t)rs.size() * 6 * FP);
    long seg_frames = 0, seg_bad = 0;
    { double mm = 0; int bad = mc_seg_fx_verify(T, &mm); fprintf(stderr, "seg fixed-point tests: %d disagreements over all window compositions, closest entropy %.3g from a cut\n", bad, mm); if (bad) return 3; }
    {   // the tabulated Seg::getprob of short windows (what k_translate_seg reads) against the function itself: every window of 100,000 random stretches
        std::vector<uint64_t> tab;
        mc_build_segtab(T.lnfac, tab);
        size_t used = 0; for (size_t i = 0; i < tab.size(); i += 2) used += tab[i] != 0;
        uint64_t rng = 0x9E3779B97F4A7C15ull; long checked = 0, badp = 0;
        for (int it = 0; it < 100000; it++) {
            uint8_t w[15];
            rng = rng * 6364136223846793005ull + 1442695040888963407ull;
            const int n = 2 + (int)((rng >> 33) % 14), alpha = 1 + (int)((rng >> 40) % 6);          // low-complexity: few distinct residues
            for (int k = 0; k < n; k++) { rng = rng * 6364136223846793005ull + 1442695040888963407ull; const int r = (int)((rng >> 35) % (unsigned)(alpha + 1)); w[k] = (uint8_t)(r == alpha ? 20 + (int)((rng >> 50) & 1) : (r * 7 + it) % 20); }
            for (int len = n; len >= 2; len--)
                for (int i = 0; i + len <= n; i++) {
                    McRgState st; st.clo = 0; st.chi = 0; st.sv = 0;
                    for (int k = 0; k < len; k++) mc_rg_add(st, w[i + k]);
                    // the kernel's form of the state: histogram of the counts, kept while the window slides in from the left end of the stretch
                    McRhState rh; rh.clo = 0; rh.chi = 0; rh.hist = 0;
                    for (int k = 0; k < len; k++) mc_rh_add(rh, w[k]);
                    for (int k = 0; k < i; k++) { mc_rh_remove(rh, w[k]); mc_rh_add(rh, w[k + len]); }
                    checked++;
                    if (rh.hist != mc_rh_of_sv(st.sv) || mc_segtab_lookup(tab.data(), rh.hist, len) != mc_seg_prob_key(mc_rg_getprob(T.lnfac, st.sv, len))) badp++;
                }
        }
        fprintf(stderr, "seg probability table: %zu pairs in %u slots; %ld windows checked, %ld differ from mc_rg_getprob\n", used, MC_SEGTAB_SLOTS, checked, badp);
        if (badp) return 3;
    }
    std::vector<int> flen(rs.size() * 6);
    // stage 1: translate + SEG
    for (size_t r = 0; r < rs.size(); r++)
        for (int f = 0; f < 6; f++) {
            uint8_t *p = &frames[(r * 6 + f) * FP];
            if ((int)rs[r].size() != read_len) { fprintf(stderr, "reads must all have the same length\n"); return 1; }
            int n = mc_translate_frame(T, (const uint8_t *)rs[r].data(), read_len, f, p);
            if (getenv("MC_SEG_PLAIN")) {
                uint8_t mask[(MC_MAXAA + 7) / 8]; double Hbuf[MC_MAXAA + 2];
                mc_seg_mask(T, p, n, mask, Hbuf);
                for (int i = 0; i < n; i++) if (mask[i >> 3] & (1 << (i & 7))) p[i] = MC_INV;
            } else if (getenv("MC_SEG_WS")) {   // double-precision workspace variant
                uint8_t comp[20], sv[24]; int16_t stk[16];
                McSegWS ws{comp, sv, stk};
                mc_seg_mask_ws(T, p, n, ws);
            } else {   // the fixed-point variant the kernel uses
                uint8_t comp[20], sv[24]; int16_t stk[16];
                McSegWS ws{comp, sv, stk};
                mc_seg_mask_fx(T.lnfac, T.seg_dout, p, n, ws);
                if (getenv("MC_CHECK_SEG")) {   // the form that computes the window flags once per frame (what the kernel runs) ...
                    std::vector<uint8_t> q2(FP, MC_INV);
                    int n3 = mc_translate_frame(T, (const uint8_t *)rs[r].data(), read_len, f, q2.data());
                    uint8_t comp2[20], sv2[24]; int16_t stk2[16];
                    McSegWS ws2{comp2, sv2, stk2};
                    mc_seg_mask_fx2(T.lnfac, T.seg_dout, q2.data(), n3, ws2);
                    if (n3 != n || memcmp(q2.data(), p, (size_t)n)) seg_bad++;
                    if (n3 >= 16) {   // ... and the register form of the composition count on a long window of the frame
                        alignas(4) uint8_t ca[20], cb[20];
                        mc_translate_frame(T, (const uint8_t *)rs[r].data(), read_len, f, q2.data());
                        const int wl = 16 + (int)((r * 7 + (size_t)f) % (size_t)(n3 - 15)), ws0 = (int)((r + (size_t)f) % (size_t)(n3 - wl + 1));
                        mc_seg_comp(q2.data() + ws0, wl, ca);
                        mc_seg_comp_rg(q2.data() + ws0, wl, cb);
                        if (memcmp(ca, cb, 20)) seg_bad++;
                    }
                    const int W3 = (n3 <= 11) ? 8 : 12;
                    if (W3 <= n3) {   // ... with the window flags from the register form of the pass
                        McBits192 a0, b0, a1, b1;
                        mc_translate_frame(T, (const uint8_t *)rs[r].data(), read_len, f, q2.data());
                        mc_seg_window_flags(T.seg_dout, q2.data(), n3, W3, comp2, a0, b0);
                        mc_seg_window_flags_rg(T.seg_dout, q2.data(), n3, W3, a1, b1);
                        if (a0.a != a1.a || a0.b != a1.b || a0.c != a1.c || b0.a != b1.a || b0.b != b1.b || b0.c != b1.c) seg_bad++;
                    }
                }
                if (getenv("MC_CHECK_SEG")) {   // ... and frame by frame against the plain restatement
                    std::vector<uint8_t> q(FP, MC_INV);
                    int n2 = mc_translate_frame(T, (const uint8_t *)rs[r].data(), read_len, f, q.data());
                    uint8_t mask[(MC_MAXAA + 7) / 8]; double Hbuf[MC_MAXAA + 2];
                    mc_seg_mask(T, q.data(), n2, mask, Hbuf);
                    for (int i = 0; i < n2; i++) if (mask[i >> 3] & (1 << (i & 7))) q[i] = MC_INV;
                    seg_frames++;
                    if (n2 != n || memcmp(q.data(), p, (size_t)n)) seg_bad++;
                }
            }
            flen[r * 6 + f] = n;
        }
    if (getenv("MC_CHECK_SEG")) {   // the kernel's shift-and-mask base codes (mc_nt_code) against the compare chains, over every byte
        int nt_bad = 0;
        for (int c = 0; c < 256; c++) nt_bad += (mc_nt_code((uint32_t)c, MC_NT_FWD_SET, MC_NT_FWD_PERM) != mc_nt_idx((uint8_t)c)) + (mc_nt_code((uint32_t)c, MC_NT_RC_SET, MC_NT_RC_PERM) != mc_nt_idx_rc((uint8_t)c));
        fprintf(stderr, "base codes: %d of 512 differ from the compare chains\n", nt_bad);
        if (nt_bad) return 3;
    }
    if (getenv("MC_CHECK_SEG")) { fprintf(stderr, "seg check: %ld frames, %ld differ from the plain restatement\n", seg_frames, seg_bad); if (seg_bad) return 3; }
    // stage 2: seed enumeration
    std::vector<McSeedTask> tasks; uint64_t lookups = 0, keyprobes = 0;
    for (size_t r = 0; r < rs.size(); r++)
        for (int f = 0; f < 6; f++) { Emit e{&tasks, (uint32_t)r, f, &X}; McSeedCount sc{0, 0, 0}; mc_enumerate_seeds(T, X, &frames[(r * 6 + f) * FP], flen[r * 6 + f], e, &sc); lookups += sc.lookups; keyprobes += sc.keyprobes; }
    if (getenv("MC_SEG_STATS")) {   // design aid: what would SEG only for the frames with a seed hit save?  
        std::vector<uint8_t> raw(FP, MC_INV);
        std::vector<McSeedTask> t2;
        long frames_n = 0, frames_hit = 0, frames_changed = 0, frames_hit_changed = 0, frames_rawhit = 0, frames_rawhit_changed = 0, raw_tasks = 0, raw_tasks_dropped = 0;
        std::vector<uint8_t> hit(rs.size() * 6, 0);
        for (const McSeedTask &t : tasks) hit[(size_t)t.read * 6 + (t.chrono >> 25)] = 1;
        for (size_t r = 0; r < rs.size(); r++)
            for (int f = 0; f < 6; f++) {
                const int n = mc_translate_frame(T, (const uint8_t *)rs[r].data(), read_len, f, raw.data());
                const uint8_t *m = &frames[(r * 6 + f) * FP];
                const bool changed = memcmp(raw.data(), m, (size_t)n) != 0;
                t2.clear();
                Emit e{&t2, (uint32_t)r, f, &X}; McSeedCount sc{0, 0, 0};
                mc_enumerate_seeds(T, X, raw.data(), n, e, &sc);
                long dropped = 0;
                for (const McSeedTask &t : t2) { const int pos = (int)((t.chrono >> 17) & 0xff), sl = (int)((t.seedlen_nkey >> 24) & 15); bool bad = false; for (int k = 0; k < sl; k++) bad |= m[pos + k] != raw[pos + k]; dropped += bad; }
                frames_n++; frames_hit += hit[r * 6 + f]; frames_changed += changed; frames_hit_changed += hit[r * 6 + f] && changed;
                frames_rawhit += !t2.empty(); frames_rawhit_changed += !t2.empty() && changed; raw_tasks += (long)t2.size(); raw_tasks_dropped += dropped;
            }
        fprintf(stderr, "seg-stats frames %ld; SEG masks something in %ld; with a seed hit %ld (masked frames), %ld (unmasked frames); with a hit AND masked %ld / %ld; seed hits on unmasked frames %ld, of them on a masked residue %ld (seed hits on masked frames %zu)\n",
                frames_n, frames_changed, frames_hit, frames_rawhit, frames_hit_changed, frames_rawhit_changed, raw_tasks, raw_tasks_dropped, tasks.size());
    }
    fprintf(stderr, "seed tasks: %zu (%.1f / read) lookups %.1f / read keyprobes %.1f / read\n", tasks.size(), (double)tasks.size() / std::max<size_t>(1, rs.size()), (double)lookups / std::max<size_t>(1, rs.size()), (double)keyprobes / std::max<size_t>(1, rs.size()));
    // stage 3: seed evaluation (+ ungapped) ; stage 4: gapped
    std::vector<McHsp> hsps; std::vector<McGapTask> gaps;
    for (const McSeedTask &t : tasks) {
        int frame = (int)(t.chrono >> 25), pos = (int)((t.chrono >> 17) & 0xff);
        McGapTask g; g.read = t.read; g.chrono = t.chrono;
        int rc = mc_eval_seed(T, X, &frames[((size_t)t.read * 6 + frame) * FP], flen[(size_t)t.read * 6 + frame], frame, pos, t.posting, (int)(t.seedlen_nkey & 0xff), (int)(t.seedlen_nkey >> 8), &g);
        if (rc == 1) { McHsp h; h.read = t.read; h.chrono = t.chrono; if (mc_make_hsp(T, read_len, frame, g, g.qfwd, g.qfwd, g.qbwd, g.qbwd, g.score, g.nmatch, g.qfwd + g.L + g.qbwd, 0, 0, &h)) hsps.push_back(h); }
        else if (rc == 2) gaps.push_back(g);
    }
    fprintf(stderr, "ungapped hsps: %zu gapped tasks: %zu\n", hsps.size(), gaps.size());
    std::vector<McGapCell> cells(2100);
    // the windowed 12-byte-cell form the kernel runs (mc_align_gapped_win) must return what the full-size form returns
    struct HostWin {   // the kernel's 8-byte cell (mc_gap_pack): what it would read back
        std::vector<uint32_t> W0, W1; uint32_t ovf = 0;
        explicit HostWin(int w) : W0(w), W1(w) {}
        void load(int c, int &h, int &d, uint32_t &ph, uint32_t &pd) const { mc_gap_unpack(W0[c], W1[c], h, d, ph, pd); }
        void store(int c, int h, int d, uint32_t ph, uint32_t pd) { ovf |= mc_gap_pack(h, d, ph, pd, W0[c], W1[c]); }
        int loadH(int c) const { return (int)(W0[c] << 20) >> 20; }
    };
    const int WIN = getenv("MC_GAP_WIN") ? atoi(getenv("MC_GAP_WIN")) : 32;
    long win_flanks = 0, win_over = 0, win_bad = 0;
    // MC_GAP_STATS (design aid): the band of every DP row - what several lanes per flank (a group of G lanes per row, G cells at a
    // time) would find to do: cells per row, rows per flank, groups of 8 / 16 per row
    const bool gap_stats = getenv("MC_GAP_STATS") != nullptr;
    long gs_flanks = 0, gs_rows = 0, gs_cells = 0, gs_g8 = 0, gs_g16 = 0, gs_hist[41] = {0}, gs_rows_hist[8] = {0}, gs_cells_long = 0;
    auto band_stats = [&](const uint8_t *s1, int st1, const uint8_t *s2, int n1, int n2) {
        HostWin ws(64);
        McGapState S;
        if (!mc_gap_begin(T, S, s1, s2, st1, n1, n2, ws, 64)) return;
        gs_flanks++;
        long rows = 0, cells = 0;
        for (;;) {
            const int js = S.jStart, je = S.jEnd < n2 ? S.jEnd : n2;
            const bool end = mc_gap_row(T, S, ws, 64);
            if (S.over) break;
            int w = (S.jEnd < je ? S.jEnd : je) - js + 1;            // (the row ran from jStart to its break column, or to the band's end; the right growth is not counted)
            if (w < 1) w = 1;
            rows++; cells += w; gs_g8 += (w + 7) / 8; gs_g16 += (w + 15) / 16; gs_hist[w > 40 ? 40 : w]++;
            if (end) break;
        }
        gs_rows += rows; gs_cells += cells;
        gs_rows_hist[rows >= 64 ? 7 : rows >= 48 ? 6 : rows >= 32 ? 5 : rows >= 24 ? 4 : rows >= 16 ? 3 : rows >= 8 ? 2 : rows >= 4 ? 1 : 0]++;
        if (rows >= 64) gs_cells_long += cells;
    };
    auto check_win = [&](const McGapResult &R, const uint8_t *s1, int st1, const uint8_t *s2, int st2, int n1, int n2) {
        if (gap_stats) band_stats(s1, st1, s2, n1, n2);
        HostWin ws(WIN);
        const McGapResult Q = mc_align_gapped_win(T, s1, st1, s2, st2, n1, n2, ws, WIN);
        win_flanks++;
        if (Q.overflow || ws.ovf) { win_over++; return; }
        if (Q.gain != R.gain || Q.c1 != R.c1 || Q.c2 != R.c2 || Q.ident != R.ident || Q.steps != R.steps || Q.runs != R.runs || Q.gapcols != R.gapcols) win_bad++;
    };
    for (const McGapTask &g : gaps) {
        int frame = (int)(g.chrono >> 25);
        const uint8_t *q = &frames[((size_t)g.read * 6 + frame) * FP]; int qlen = flen[(size_t)g.read * 6 + frame];
        const uint8_t *d = X.res + X.off[g.sidx]; int dlen = (int)(X.off[g.sidx + 1] - X.off[g.sidx]);
        int score = g.score, nmatch = g.nmatch, qfwd = g.qfwd, dfwd = g.qfwd, qbwd = g.qbwd, dbwd = g.qbwd;
        int alnlen = g.qfwd + g.L + g.qbwd, gapopens = 0, gaptotal = 0;
        int qend = qfwd + g.qp + g.L, dend = dfwd + g.dp + g.L, dright = dlen - dend, qright = qlen - qend;
        if (dright > 2 && qright > 2) {
            McGapResult R = mc_align_gapped(T, q + qend, 1, d + dend, 1, qright, dright, cells.data(), (int)cells.size());
            check_win(R, q + qend, 1, d + dend, 1, qright, dright);
            if (R.gain > 0) { score += R.gain; nmatch += R.ident; qfwd += R.c1; dfwd += R.c2; alnlen += R.steps; gapopens += R.runs; gaptotal += R.gapcols; }
        }
        int dleft = g.dp - dbwd, qleft = g.qp - qbwd;
        if (dleft > 2 && qleft > 2) {
            McGapResult R = mc_align_gapped(T, q + qleft - 1, -1, d + dleft - 1, -1, qleft, dleft, cells.data(), (int)cells.size());
            check_win(R, q + qleft - 1, -1, d + dleft - 1, -1, qleft, dleft);
            if (R.gain > 0) { score += R.gain; nmatch += R.ident; qbwd += R.c1; dbwd += R.c2; alnlen += R.steps; gapopens += R.runs; gaptotal += R.gapcols; }
        }
        McHsp h; h.read = g.read; h.chrono = g.chrono;
        if (mc_make_hsp(T, read_len, frame, g, qfwd, dfwd, qbwd, dbwd, score, nmatch, alnlen, gapopens, gaptotal, &h)) hsps.push_back(h);
    }
    if (const char *pre = getenv("MC_DUMP_STAGES")) {   // what every stage made, for the per-stage parity test of the kernels (tests/test_gpu_parity.py)
        auto dump = [&](const char *name, const void *p, size_t bytes) { const std::string fn = std::string(pre) + name; FILE *d = fopen(fn.c_str(), "wb"); if (!d || (bytes && fwrite(p, 1, bytes, d) != bytes)) { fprintf(stderr, "cannot write %s\n", fn.c_str()); exit(2); } fclose(d); };
        dump(".frames", frames.data(), frames.size());                 // rows of MC_MAXAA + 2 bytes
        dump(".tasks", tasks.data(), tasks.size() * sizeof(McSeedTask));
        dump(".gaps", gaps.data(), gaps.size() * sizeof(McGapTask));
        dump(".hsps", hsps.data(), hsps.size() * sizeof(McHsp));       // (ungapped and gapped, before any ordering)
        fprintf(stderr, "stages dumped: FP %d, %zu B per task, %zu per gap task, %zu per HSP\n", FP, sizeof(McSeedTask), sizeof(McGapTask), sizeof(McHsp));
    }
    if (gap_stats) {
        fprintf(stderr, "gap-stats: %ld flanks (every task's two, not deduplicated), %ld rows, %ld cells: %.1f rows per flank, %.2f cells per row; groups per row of 8 lanes %.2f (%.0f %% of the lanes busy), of 16 lanes %.2f (%.0f %%); cells in flanks of >= 64 rows: %.1f %%\n",
                gs_flanks, gs_rows, gs_cells, (double)gs_rows / (double)(gs_flanks ? gs_flanks : 1), (double)gs_cells / (double)(gs_rows ? gs_rows : 1),
                (double)gs_g8 / (double)(gs_rows ? gs_rows : 1), 100.0 * (double)gs_cells / (8.0 * (double)(gs_g8 ? gs_g8 : 1)), (double)gs_g16 / (double)(gs_rows ? gs_rows : 1), 100.0 * (double)gs_cells / (16.0 * (double)(gs_g16 ? gs_g16 : 1)),
                100.0 * (double)gs_cells_long / (double)(gs_cells ? gs_cells : 1));
        fprintf(stderr, "gap-stats: rows per flank < 4 / 4-7 / 8-15 / 16-23 / 24-31 / 32-47 / 48-63 / >= 64:");
        for (int k = 0; k < 8; k++) fprintf(stderr, " %ld", gs_rows_hist[k]);
        fprintf(stderr, "\ngap-stats: cells per row 1 .. 40+:");
        for (int k = 1; k <= 40; k++) fprintf(stderr, " %ld", gs_hist[k]);
        fprintf(stderr, "\n");
    }
    fprintf(stderr, "gapped window check (W = %d): %ld flanks, %ld leave the window, %ld differ from the full-size form\n", WIN, win_flanks, win_over, win_bad);
    if (win_bad) return 3;
    // stage 5: sort by (read, subject, chrono)
    std::sort(hsps.begin(), hsps.end(), [](const McHsp &a, const McHsp &b) { if (a.read != b.read) return a.read < b.read; if (a.sidx != b.sidx) return a.sidx < b.sidx; return a.chrono < b.chrono; });
    fprintf(stderr, "hsps kept: %zu\n", hsps.size());
    if (getenv("MC_HSP_STATS")) {   // how many reads could print anything at all, by criteria that need no sorted order (design aid for the HSP binning)
        long reads = 0, with_low = 0, with_pair = 0, with_diff = 0, with_rows_possible = 0, h_all = 0, h_low = 0, h_pair = 0, h_diff = 0, dup = 0;
        for (size_t a = 0; a < hsps.size();) {
            size_t b = a; while (b < hsps.size() && hsps[b].read == hsps[a].read) b++;
            bool low = false, pair = false, diff = false;
            for (size_t i = a; i < b; i++) {
                if (hsps[i].loge < T.loge_thr) low = true;
                if (i > a && hsps[i].sidx == hsps[i - 1].sidx) {
                    pair = true;
                    const McHsp &p = hsps[i - 1], &h = hsps[i];
                    if (!(p.frame == h.frame && p.qaas == h.qaas && p.ds == h.ds && p.qaae == h.qaae && p.de == h.de)) diff = true; else dup++;
                }
            }
            { const long n = (long)(b - a); static long hist[8]; static long hh[8]; const long lim[8] = {16, 32, 64, 128, 512, 2048, 8192, 1L << 40};
              for (int k = 0; k < 8; k++) if (n <= lim[k]) { hist[k]++; hh[k] += n; break; }
              if (b == hsps.size()) { fprintf(stderr, "hsp-stats segment sizes (<=16, 32, 64, 128, 512, 2048, 8192, more): reads"); for (int k = 0; k < 8; k++) fprintf(stderr, " %ld", hist[k]); fprintf(stderr, "; HSPs"); for (int k = 0; k < 8; k++) fprintf(stderr, " %ld", hh[k]); fprintf(stderr, "\n"); } }
            reads++; h_all += (long)(b - a);
            if (low) { with_low++; h_low += (long)(b - a); }
            if (low || pair) { with_pair++; h_pair += (long)(b - a); }
            if (low || diff) { with_diff++; h_diff += (long)(b - a); }
            a = b;
        }
        fprintf(stderr, "hsp-stats reads with HSPs %ld (HSPs %ld, duplicates of the HSP in front %ld); with an HSP below the threshold %ld (their HSPs %ld); ... or two HSPs on one subject %ld (%ld); ... or two DIFFERENT HSPs on one subject %ld (%ld)\n",
                reads, h_all, dup, with_low, h_low, with_pair, h_pair, with_diff, h_diff);
    }
    // stage 6: per-read finishing
    FILE *o = fopen(argv[3], "w");
    std::vector<McHsp> v, tmp; std::vector<McRow> rows(MC_MAX_M8); std::vector<double> kr(MC_MAX_M8); std::vector<McSortItem> items;
    for (size_t a = 0; a < hsps.size();) {
        size_t b = a; while (b < hsps.size() && hsps[b].read == hsps[a].read) b++;
        int n = (int)(b - a); v.resize(n); tmp.resize(2 * n); items.resize(n);
        McBestHit best;
        int nr = mc_finish_read(T, X, P, fam.data(), (int)hsps[a].read, &hsps[a], n, v.data(), tmp.data(), rows.data(), kr.data(), items.data(), &best);
        for (int i = 0; i < nr; i++) {
            const McRow &r = rows[i];
            fprintf(o, "%s\t%s\t%g\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%g\t%g\n", rn[r.query].c_str(), H.names[r.subject].c_str(), r.ident, r.alnlen, r.mismatch, r.gapopen, r.qstart, r.qend, r.sstart, r.send, r.loge, r.bits);
        }
        a = b;
    }
    fclose(o);
    return 0;
}
