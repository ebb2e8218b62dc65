// mc_hip.hip - libmcensus_hip.so for gfx950: the handle, the five-stage pipeline of a batch, the streaming forms and the C ABI
// (include/mcensus.h names what each entry point replaces in the reference).  One translation unit; the kernels live in
//   mc_hip_common.h    includes, counters, wave-level helpers
//   k_translate_seg.h  A1  six-frame translation + SEG: a wave per 10 reads, the trimming search dealt out over the wave
//   k_enumerate.h      A2  seed enumeration + index probes: a wave per read, a state machine over per-wave LDS queues that persist across the reads of a chunk (k_enumerate_q)
//   k_eval_seeds.h     A3  seed gate, growth, ungapped X-drop: persistent waves, gate and extension as two phases of a wave
//   k_gapped.h         B   gapped X-drop: one DP per distinct segment, a lane per flank, DP rows packed in LDS
//   k_order.h          C   HSPs binned per read (no global sort), ordered and stacked for the reads that can print
//   k_finish.h         D   sum statistics, std::sort / heap sort replayed, 500-row cap, classification; rows in m8 order
//   k_grid.h               the training workflow's grid classification
// and the per-thread algorithms they share with the test-only emulation in mc_core.h / mc_finish.h / mc_index.h.
// Stage E copies rows and best hits to pinned host memory.  mc_run_range() issues the stages of one range; run_stream() feeds
// ranges from a host-side source (mc_search, mc_search_files, mc_search_files_multi) with upload and search overlapped.
#include "mc_hip_common.h"
#include "k_translate_seg.h"
#include "k_enumerate.h"
#include "k_eval_seeds.h"
#include "k_gapped.h"
#include "k_order.h"
#include "k_finish.h"
#include "k_grid.h"

// ------------------------------------------------------------------------------------------------
// handle
// ------------------------------------------------------------------------------------------------
// Everything the range in flight needs: streams, events, the pools of the stages, the pinned mirrors of the device counters.  (Rounds 2 - 3
// had two of them per handle for a range run as two interleaved halves; what overlaps now is the host's work on the results of one
// range with the front of the next, and that needs no second set of pools - range_begin.)
#define MC_NCTX 1
struct McCtx {
    hipStream_t stream = nullptr, side = nullptr, side2 = nullptr;   // the pipeline of a range, and two side streams of the ordering / finishing kernels
    hipEvent_t ev[8] = {}, ev_fork = nullptr, ev_join = nullptr, ev_join2 = nullptr, ev_part[16] = {}, ev_en[16] = {};
    hipStream_t tr_stream = nullptr;                               // the translation of the next part of a range beside the seed search of this one (stage_a, MC_A_PARTS)
    int64_t cap_reads = 0;
    uint32_t cap_tasks = 0, cap_gaps = 0, cap_hsps = 0, cap_rows = 0;
    uint8_t *d_frames = nullptr, *d_frames_base = nullptr;   // (64 bytes of room in front: k_eval_seeds reads 8 bytes at a time backwards from a seed)
    unsigned long long *d_stats = nullptr;
    McSeedTask *d_tasks = nullptr; McGapTask *d_gaps = nullptr; McHsp *d_hsps = nullptr, *d_v = nullptr, *d_tmp = nullptr;
    uint64_t *d_k64 = nullptr, *d_hkeys = nullptr, *d_hplace = nullptr, *d_places = nullptr; uint32_t *d_idx = nullptr, *d_idxo = nullptr, *d_heads = nullptr, *d_scan = nullptr, *d_gsz = nullptr, *d_nv = nullptr; uint32_t *d_ghist = nullptr;
    uint32_t *d_counters = nullptr;
    McRow *d_rows = nullptr; uint32_t *d_nrow = nullptr, *d_rowoff = nullptr; McBestHit *d_best = nullptr, *d_bestof = nullptr; uint8_t *d_low = nullptr, *d_cand = nullptr;
    McGapCell *d_gws_full = nullptr; uint32_t *d_retry = nullptr, *d_retry2 = nullptr; int gap_threads_full = 0;
    unsigned long long *d_gtab = nullptr; uint32_t gtab_slots = 0; uint32_t *d_gleader = nullptr; McFlankOut *d_fout = nullptr;
    // pinned host mirrors
    uint32_t *h_c = nullptr; unsigned long long *h_stats = nullptr; McBestHit *h_best = nullptr; size_t h_best_cap = 0;
    // the range being processed
    const uint8_t *reads = nullptr; int64_t n = 0, first_read_id = 0;
    uint32_t ntasks = 0, ngaps = 0, gpad = 0, nh = 0, nh_all = 0, nheads = 0, nrows = 0, nbest = 0, nsegs = 0;
    bool busy = false;                                             // a range has been begun and not ended
};

struct mc_handle {
    McHostIndex H;
    std::vector<int32_t> fam;
    int nfam = 0, device = 0;
    // device index + tables
    uint8_t *d_res = nullptr, *d_res_base = nullptr; uint32_t *d_off = nullptr, *d_bstart = nullptr, *d_post = nullptr; unsigned long long *d_post8 = nullptr; uint16_t *d_keys = nullptr; int32_t *d_fam = nullptr;
    McTables *d_T = nullptr; McClassPars *d_P = nullptr;
    McTables hT; McClassPars hP;
    int read_len = 0, FP = 0; bool run_set = false;
    uint32_t *d_bitmap = nullptr;
    McBucketRec *d_rec = nullptr;
    uint32_t *d_filt = nullptr, *d_wild = nullptr, *d_pair = nullptr; uint64_t *d_segtab = nullptr; unsigned long long *d_rt = nullptr;
    bool fast_enum = false;
    bool count_traffic = false;
    int pipe_nout = 0;                    // mc_range_begin / mc_range_end: 1 while a range has been begun and not ended
    bool keep_rows = true;                // mc_search / mc_search_files hand out the m8 rows (mc_set_keep_rows)
    bool best_only = false;               // only the reads that can be classified are ranked; no rows (mc_set_best_hits_only)
    uint8_t *stage_pin[2] = {}, *stage_dev[2] = {}; size_t stage_bytes = 0; hipStream_t copy_stream = nullptr;   // run_stream
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
    hipStream_t side = nullptr, side2 = nullptr;                    // (McCtx::side, side2)
    hipStream_t rows_stream = nullptr; hipEvent_t ev_rows = nullptr; bool rows_pending = false, rows_ever = false;
    std::vector<mc_row> all_rows, split_rows;                       // accumulated over the batches of a stream / over the halves of a range that overflowed
    const mc_row *res_rows = nullptr; int64_t n_res_rows = 0;
    std::vector<mc_best_hit> best; mc_stats stats;
    McCtx *best_from = nullptr; uint32_t best_count = 0;           // the context whose best hits (pinned, unordered) are those of the last range, and how many: best_materialize
};

static McIndex dev_index(const mc_handle *h)
{
    McIndex X; X.res = h->d_res; X.off = h->d_off; X.bstart = h->d_bstart; X.post = h->d_post; X.post8 = h->d_post8; X.keys = h->d_keys; X.rec = h->d_rec; X.filt = h->d_filt; X.wild = h->d_wild; X.pair = h->d_pair; X.rt = h->d_rt; X.rt_mask = h->H.rt_mask; X.nseq = h->H.nseq;
    return X;
}

template <class Tp> static int dalloc(Tp **p, size_t n)
{
    if (*p) { (void)hipFree(*p); *p = nullptr; }
    HIPCK(hipMalloc((void **)p, n * sizeof(Tp)));
    return 0;
}

// (Rounds 3 - 4 had a load-time constructor here that exported GPU_MAX_HW_QUEUES=8: a process-wide side effect on every other HIP
// user of the host application, dependent on load order, and a setenv() that races with getenv() in other threads - ADVICE r04.  The
// library no longer touches the environment; the entry points that own their process - scripts/*, bench.py, the tests - set the
// variable before anything initialises HIP, and INTEGRATION.md section 3 tells an embedding application to do the same.)

extern "C" int mc_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static void ctx_free(McCtx &c)
{
    void *ptrs[] = {c.d_frames_base, c.d_tasks, c.d_gaps, c.d_hsps, c.d_v, c.d_tmp, c.d_k64, c.d_hkeys, c.d_hplace, c.d_places, c.d_idx, c.d_idxo, c.d_heads, c.d_scan, c.d_gsz, c.d_nv, c.d_ghist, c.d_counters, c.d_rows,
                    c.d_nrow, c.d_rowoff, c.d_best, c.d_bestof, c.d_low, c.d_cand, c.d_gws_full, c.d_retry, c.d_retry2, c.d_gtab, c.d_gleader, c.d_fout, c.d_stats};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    for (void *p : {(void *)c.h_c, (void *)c.h_stats, (void *)c.h_best}) if (p) (void)hipHostFree(p);
    for (auto &e : c.ev) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : {c.ev_fork, c.ev_join, c.ev_join2}) if (e) (void)hipEventDestroy(e);
    for (auto &e : c.ev_part) if (e) (void)hipEventDestroy(e);
    for (auto &e : c.ev_en) if (e) (void)hipEventDestroy(e);
    if (c.tr_stream) (void)hipStreamDestroy(c.tr_stream);
    if (c.stream) (void)hipStreamDestroy(c.stream);                // (side, side2: the handle's)
    c = McCtx();
}

extern "C" void mc_close(mc_handle *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    void *ptrs[] = {h->d_res_base, h->d_off, h->d_bstart, h->d_post, h->d_post8, h->d_keys, h->d_fam, h->d_T, h->d_P, h->d_reads, h->d_bitmap, h->d_rec, h->d_filt, h->d_wild, h->d_pair, h->d_rt, h->d_segtab};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    for (McCtx &c : h->ctx) ctx_free(c);
    for (int k = 0; k < 2; k++) { if (h->stage_pin[k]) (void)hipHostFree(h->stage_pin[k]); if (h->stage_dev[k]) (void)hipFree(h->stage_dev[k]); }
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    for (hipStream_t q : {h->side, h->side2}) if (q) (void)hipStreamDestroy(q);
    if (h->rows_stream) { (void)hipStreamSynchronize(h->rows_stream); (void)hipStreamDestroy(h->rows_stream); }
    if (h->ev_rows) (void)hipEventDestroy(h->ev_rows);
    for (mc_row *p : h->pin_slot) if (p) (void)hipHostFree(p);
    delete h;
}

// Host arrays -> device through two pinned bounce buffers (a memcpy into one while the other travels).  hipMemcpy from pageable
// memory pins the caller's pages on the way: the 110 MB of the index took 80 - 150 ms of the engine's 120 - 190 ms warm-cache
// open that way (the reference's default use is ONE run per process: microbe_census.py:375) - this takes ~25.
struct McUploader {
    static constexpr size_t CH = (size_t)8 << 20;
    uint8_t *pin[2] = {nullptr, nullptr}; hipEvent_t ev[2] = {nullptr, nullptr}; hipStream_t st = nullptr; int k = 0; bool used[2] = {false, false};
    int init()
    {
        HIPCK(hipStreamCreate(&st));
        for (int i = 0; i < 2; i++) { HIPCK(hipHostMalloc((void **)&pin[i], CH, hipHostMallocDefault)); HIPCK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming)); }
        return 0;
    }
    int put(void *dst, const void *src, size_t bytes)
    {
        for (size_t at = 0; at < bytes; at += CH) {
            const size_t n = std::min(CH, bytes - at);
            if (used[k]) HIPCK(hipEventSynchronize(ev[k]));
            memcpy(pin[k], (const uint8_t *)src + at, n);
            HIPCK(hipMemcpyAsync((uint8_t *)dst + at, pin[k], n, hipMemcpyHostToDevice, st));
            HIPCK(hipEventRecord(ev[k], st)); used[k] = true;
            k ^= 1;
        }
        return 0;
    }
    int finish() { if (st) HIPCK(hipStreamSynchronize(st)); return 0; }
    ~McUploader()
    {
        if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
        for (int i = 0; i < 2; i++) { if (pin[i]) (void)hipHostFree(pin[i]); if (ev[i]) (void)hipEventDestroy(ev[i]); }
    }
};

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
    // Streams: one for the pipeline of a range, two side streams for the ordering / finishing kernels of the longest reads
    // (the handle's), one for the rows on their way to the host, one for the
    // uploads of the streaming calls.  HIP multiplexes its streams onto GPU_MAX_HW_QUEUES hardware queues (4 unless the environment
    // says otherwise) and streams that share one wait for each other: with nine streams the front of a range could land behind the
    // 5 ms copy of the rows of the range before (measured: 51.2 instead of 53.6 M reads/s) - hence few streams, and GPU_MAX_HW_QUEUES=8
    // exported by the entry points (microbecensus_amd.configure_process_env(); never by this library).
    HIPCK(hipStreamCreate(&h->side)); HIPCK(hipStreamCreate(&h->side2));
    for (McCtx &c : h->ctx) {
        HIPCK(hipStreamCreate(&c.stream)); c.side = h->side; c.side2 = h->side2;
        for (auto &e : c.ev) HIPCK(hipEventCreate(&e));
        HIPCK(hipEventCreateWithFlags(&c.ev_fork, hipEventDisableTiming)); HIPCK(hipEventCreateWithFlags(&c.ev_join, hipEventDisableTiming));
        HIPCK(hipEventCreateWithFlags(&c.ev_join2, hipEventDisableTiming));
        if (dalloc(&c.d_counters, C_N) || dalloc(&c.d_stats, S_N) || dalloc(&c.d_ghist, (size_t)MC_GS_BINS)) return -1;
        HIPCK(hipHostMalloc((void **)&c.h_c, sizeof(uint32_t) * C_N, hipHostMallocDefault));
        HIPCK(hipHostMalloc((void **)&c.h_stats, sizeof(unsigned long long) * S_N, hipHostMallocDefault));
    }
    HIPCK(hipStreamCreate(&h->rows_stream)); HIPCK(hipEventCreateWithFlags(&h->ev_rows, hipEventDisableTiming));
    const McHostIndex &H = h->H;
    if (H.res.size() >= MC_TASK_ABS_LIMIT) { g_err = "marker database too large: more than 16 M residues (MC_TASK_W3)"; return -1; }
    if (dalloc(&h->d_res_base, H.res.size() + 128) || dalloc(&h->d_off, H.off.size()) || dalloc(&h->d_bstart, H.bstart.size()) || dalloc(&h->d_post, H.post.size() + 1) || dalloc(&h->d_post8, (H.post.size() + 1) * MC_POST_WORDS) ||
        dalloc(&h->d_keys, H.keys.size()) || dalloc(&h->d_fam, (size_t)nseq) || dalloc(&h->d_T, 1) || dalloc(&h->d_P, 1)) return -1;
    HIPCK(hipMemset(h->d_res_base, MC_INV, H.res.size() + 128));
    h->d_res = h->d_res_base + 64;                                 // (k_gapped_lds reads 16 bytes at a time around a flank's first residues)
    if (dalloc(&h->d_bitmap, H.bitmap.size()) || dalloc(&h->d_filt, H.filt.size()) || dalloc(&h->d_wild, H.wild.size()) || dalloc(&h->d_pair, H.pair.size()) || dalloc(&h->d_rt, H.rt.size())) return -1;
    if (!H.rec.empty() && dalloc(&h->d_rec, H.rec.size())) return -1;
    {
        McUploader up;
        if (up.init() || up.put(h->d_res, H.res.data(), H.res.size()) || up.put(h->d_off, H.off.data(), H.off.size() * 4) || up.put(h->d_bstart, H.bstart.data(), H.bstart.size() * 4) ||
            up.put(h->d_post, H.post.data(), H.post.size() * 4) || up.put(h->d_keys, H.keys.data(), H.keys.size() * 2) || up.put(h->d_fam, h->fam.data(), (size_t)nseq * 4) ||
            up.put(h->d_bitmap, H.bitmap.data(), H.bitmap.size() * 4) || up.put(h->d_pair, H.pair.data(), H.pair.size() * 4) || up.put(h->d_wild, H.wild.data(), H.wild.size() * 4) ||
            up.put(h->d_rt, H.rt.data(), H.rt.size() * 8) || up.put(h->d_filt, H.filt.data(), H.filt.size() * 4) ||
            (!H.rec.empty() && up.put(h->d_rec, H.rec.data(), H.rec.size() * sizeof(McBucketRec))) || up.finish()) return -1;
    }
    if (H.nseq > 32767) { g_err = "marker database too large: more than 32,767 sequences (MC_POST8)"; return -1; }
    k_post8<<<dim3((unsigned)((H.post.size() + 255) / 256)), dim3(256)>>>(h->d_post, h->d_off, h->d_res, (uint32_t)H.post.size(), h->d_post8);
    HIPCK(hipDeviceSynchronize());
    MC_OT("  index upload", t0);
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

// Host only (no GPU): does the index cache give back what was built?  Builds the index, writes it to <dir>, reads it back and
// compares every array; then damages one byte of the file and expects the load to refuse it.  0 = all of that held.
extern "C" int mc_index_cache_check(const char *const *names, const char *const *seqs, int32_t nseq, const char *dir)
{
    McHostIndex A, B, C2;
    std::string err;
    if (!dir || !mc_build_index(A, names, seqs, nseq, err)) { g_err = err.empty() ? "bad argument" : err; return -1; }
    const uint64_t ih = mc_ixc_input_hash(names, seqs, nseq);
    char nm[64]; snprintf(nm, sizeof nm, "/index_%016llx.mcix", (unsigned long long)ih);
    const std::string path = std::string(dir) + nm;
    if (!mc_index_save(A, ih, path.c_str())) { g_err = "cannot write " + path; return -1; }
    if (!mc_index_load(B, ih, nseq, path.c_str())) { g_err = "the file just written was not accepted"; return 1; }
    if (!(A.names == B.names && A.res_code == B.res_code && A.res == B.res && A.off == B.off && A.bstart == B.bstart && A.post == B.post && A.keys == B.keys && A.bitmap == B.bitmap &&
          A.filt == B.filt && A.wild == B.wild && A.pair == B.pair && A.rt == B.rt && A.rec.size() == B.rec.size() &&
          (A.rec.empty() || memcmp(A.rec.data(), B.rec.data(), A.rec.size() * sizeof(McBucketRec)) == 0) && A.rt_mask == B.rt_mask && A.max_bucket == B.max_bucket &&
          A.freq_thr == B.freq_thr && A.nres == B.nres && A.nseq == B.nseq && memcmp(A.letter_p, B.letter_p, sizeof A.letter_p) == 0)) { g_err = "the index read back differs from the one built"; return 2; }
    if (mc_index_load(C2, ih ^ 1, nseq, path.c_str())) { g_err = "a file of other sequences was accepted"; return 3; }
    if (!mc_index_matches_input(B, names, seqs, nseq)) { g_err = "the index read back does not pass the input comparison"; return 6; }
    {   // what the checksum cannot see: a well-formed file that is not the index of these sequences (a hash collision, a stale layout)
        McHostIndex D = B;
        if (D.nres > 0) { D.res[(size_t)D.nres / 2] ^= 1; if (mc_index_matches_input(D, names, seqs, nseq)) { g_err = "an index with another residue was accepted"; return 7; } D.res[(size_t)D.nres / 2] ^= 1; }
        if (!D.post.empty()) { const uint32_t keep = D.post[D.post.size() / 2]; D.post[D.post.size() / 2] = ((uint32_t)nseq << 11); if (mc_index_matches_input(D, names, seqs, nseq)) { g_err = "a posting outside the database was accepted"; return 8; } D.post[D.post.size() / 2] = keep; }
        if (!D.rec.empty()) { D.rec[D.rec.size() / 3].start += 1; if (mc_index_matches_input(D, names, seqs, nseq)) { g_err = "a bucket record outside its bucket was accepted"; return 9; } D.rec[D.rec.size() / 3].start -= 1; }
        if (nseq > 1) { std::vector<const char *> nm2(names, names + nseq); std::swap(nm2[0], nm2[1]); if (mc_index_matches_input(D, nm2.data(), seqs, nseq)) { g_err = "other marker names were accepted"; return 10; } }
    }
    {   // one byte of the payload flipped: the checksum must notice
        FILE *f = fopen(path.c_str(), "r+b");
        if (!f) { g_err = "cannot reopen " + path; return -1; }
        fseek(f, 0, SEEK_END); const long sz = ftell(f);
        fseek(f, sz / 2, SEEK_SET); int c = fgetc(f); fseek(f, sz / 2, SEEK_SET); fputc(c ^ 0x40, f); fclose(f);
        if (mc_index_load(C2, ih, nseq, path.c_str())) { g_err = "a damaged file was accepted"; return 4; }
        f = fopen(path.c_str(), "r+b"); fseek(f, sz / 2, SEEK_SET); fputc(c, f); fclose(f);
        if (truncate(path.c_str(), sz - 9) != 0 || mc_index_load(C2, ih, nseq, path.c_str())) { g_err = "a truncated file was accepted"; return 5; }
    }
    remove(path.c_str());
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
        if (loaded && !mc_index_matches_input(h->H, names, seqs, nseq)) { loaded = false; h->H = McHostIndex(); }   // (not the index of THESE sequences, or offsets out of range: rebuilt)
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

static void best_materialize(mc_handle *h);
static int ensure_capacity(mc_handle *h, McCtx &c, int64_t nreads)
{
    if (nreads <= c.cap_reads) return 0;
    if (h->best_from == &c) best_materialize(h);                   // (the best hits of the range before still lie in the pinned buffer that is about to be replaced)
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

// The pipeline of a range, in five stages.  Each stage only ISSUES work on the range's stream and ends with an asynchronous copy
// of the device counters into pinned host memory; the next stage starts by waiting for that copy (stage_wait) and sizes its
// launches from it.
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
    // The front in PARTS (MC_A_PARTS = P > 1): the translation of part p + 1 on a side stream beside the seed search of
    // part p.  The translation is bound by VALU issue with the memory system idle, the seed search by scattered lines with two in five issue
    // slots idle; side by side as equals they took from each other what they gained (a translation wave keeps its SIMD's
    // issue slots busy and the seed waves beside it stand still) - so the seed kernel's waves run at a higher issue priority (MC_EN_PRIO,
    // s_setprio) and the translation takes the slots they leave.
    static const int a_parts = getenv("MC_A_PARTS") ? std::max(1, std::min(16, atoi(getenv("MC_A_PARTS")))) : 1;
    static const int en_prio = getenv("MC_EN_PRIO") ? std::max(0, std::min(3, atoi(getenv("MC_EN_PRIO")))) : 0;
    const bool in_parts = a_parts > 1 && h->fast_enum && !h->count_traffic && n >= (int64_t)a_parts * 4096;
    int64_t part_n = n;
    if (in_parts) part_n = (((n + a_parts - 1) / a_parts) + 1023) / 1024 * 1024;
    auto translate = [&](hipStream_t s2, int64_t off, int64_t cnt) -> int {
        if (staged) {
            if (lds > 48 * 1024) HIPCK(hipFuncSetAttribute((const void *)k_translate_seg<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            k_translate_seg<true><<<dim3((unsigned)((cnt + MC_TS_READS - 1) / MC_TS_READS)), dim3(MC_TS_THREADS), lds, s2>>>(h->d_T, c.reads + off * L, L, cnt, c.d_frames + off * 6 * FP, FP, h->d_segtab);
        } else {
            if (lds > 48 * 1024) HIPCK(hipFuncSetAttribute((const void *)k_translate_seg<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            k_translate_seg<false><<<dim3((unsigned)((cnt + MC_TS_READS - 1) / MC_TS_READS)), dim3(MC_TS_THREADS), lds, s2>>>(h->d_T, c.reads + off * L, L, cnt, c.d_frames + off * 6 * FP, FP, h->d_segtab);
        }
        return 0;
    };
    if (!in_parts) { if (translate(st, 0, n)) return -1; }
    else {
        // T0 | E0 + T1 | E1 + T2 | ...: the translation of part p + 1 is released together with the seed search of part p (it waits for the
        // search of part p - 1; left to itself the side stream would run ALL translations first - their small workgroups take every slot
        // that frees up before a seed workgroup of 8 waves fits).  A stream of the lowest priority: the seed kernel's workgroups are placed first.
        if (!c.tr_stream) { int lo = 0, hi = 0; HIPCK(hipDeviceGetStreamPriorityRange(&lo, &hi)); HIPCK(hipStreamCreateWithPriority(&c.tr_stream, hipStreamNonBlocking, lo)); }
        for (int p = 0; (int64_t)p * part_n < n; p++) {
            if (!c.ev_part[p]) HIPCK(hipEventCreateWithFlags(&c.ev_part[p], hipEventDisableTiming));
            if (!c.ev_en[p]) HIPCK(hipEventCreateWithFlags(&c.ev_en[p], hipEventDisableTiming));
        }
        HIPCK(hipEventRecord(c.ev_fork, st));
        HIPCK(hipStreamWaitEvent(c.tr_stream, c.ev_fork, 0));
        if (translate(c.tr_stream, 0, std::min(part_n, n))) return -1;
        HIPCK(hipEventRecord(c.ev_part[0], c.tr_stream));
        HIPCK(hipStreamWaitEvent(st, c.ev_part[0], 0));               // (what the stage's first timer sees of the translation: its first part)
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
        // k_enumerate_q (round 5: queues that persist across the reads of a chunk) is the kernel of the product path; the counting form
        // (mc_set_counting: what the reference would read) is k_enumerate_count (rounds 2 - 4's one-wave-per-read kernel without filters).
        const bool enq = !h->count_traffic;
        const size_t per_wave = enq ? MC_ENQ_WAVE_LDS(FP, L) : sizeof(McEnWave) + MC_EN_WAVE_LDS(FP, L);
        // Launch shape.  k_enumerate_q: SIXTEEN waves per CU (2 workgroups of 8) where the LDS holds them - the kernel is bound by the
        // scattered lines its CU's vector L1 has to fetch, not by issue or latency, and more resident waves only thrash
        // that cache: per 1 M reads of 100 / 150 / 300 bp 24 (20 at 300 bp) waves 3.85 / 6.23 / 13.29 ms, 16 waves 3.77 / 6.11 / 13.21.
        // The counting form (rounds 2 - 4's kernel, issue bound): as many as fit, up to 24 (16 waves 6.77 ms, 20: 6.45, 24: 6.39).
        int waves = 0, bpc = 1;
        {
            static const int shapes_q[][2] = {{8, 2}, {4, 4}, {16, 1}, {12, 1}, {4, 3}, {8, 1}, {4, 2}, {4, 1}};
            static const int shapes_c[][2] = {{12, 2}, {8, 3}, {4, 6}, {4, 5}, {16, 1}, {8, 2}, {4, 4}, {12, 1}, {4, 3}, {8, 1}, {4, 2}, {4, 1}};
            if (enq) { for (const auto &sh : shapes_q) if (!waves && (size_t)sh[1] * (64 + sh[0] * per_wave) <= 160 * 1024) { waves = sh[0]; bpc = sh[1]; } }
            else for (const auto &sh : shapes_c) if (!waves && (size_t)sh[1] * (64 + sh[0] * per_wave) <= 160 * 1024) { waves = sh[0]; bpc = sh[1]; }
        }
        if (!waves) { g_err = "reads too long for the seed kernel's LDS layout"; return -1; }
        if (const char *e = getenv("MC_EN_SHAPE")) { int a = 0, b = 0; if (sscanf(e, "%d,%d", &a, &b) == 2 && (a == 16 || a == 12 || a == 8 || a == 4) && b >= 1 && (size_t)b * (64 + a * per_wave) <= 160 * 1024) { waves = a; bpc = b; } }   // (experiments)
        const size_t lds2 = 64 + waves * per_wave;
#ifdef MC_EN_FRONT_ONLY   /* measurement build: the lists the front writes its wildcard asks to */
        static uint32_t *fo_items = nullptr, *fo_cursor = nullptr;
        const uint32_t fo_cap = 80u << 20;                          /* slots per list: 8 x 80 M x 12 B = 7.7 GB */
        if (!fo_items) { HIPCK(hipMalloc((void **)&fo_items, (size_t)8 * fo_cap * 12)); HIPCK(hipMalloc((void **)&fo_cursor, 8 * 32 * 4)); }
        HIPCK(hipMemsetAsync(fo_cursor, 0, 8 * 32 * 4, st));
#define MC_FO_ARGS , fo_items, fo_cap, fo_cursor
#else
#define MC_FO_ARGS
#endif
#define MC_LAUNCH_EN(KERNEL, WV)                                                                                                                   \
    do {                                                                                                                                           \
        HIPCK(hipFuncSetAttribute((const void *)KERNEL<WV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));                               \
        for (int p_ = 0; (int64_t)p_ * part_n < n; p_++) {                                                                                         \
            const int64_t off_ = (int64_t)p_ * part_n, cnt_ = std::min(part_n, n - off_);                                                          \
            const int blocks_ = (int)std::min<int64_t>((int64_t)256 * bpc, (cnt_ + WV - 1) / WV);                                                  \
            if (in_parts) {                                                                                                                        \
                const int64_t off2_ = off_ + part_n;                                                                                               \
                if (off2_ < n) {                                                                                                                   \
                    if (p_ >= 1) HIPCK(hipStreamWaitEvent(c.tr_stream, c.ev_en[p_ - 1], 0));                                                       \
                    if (translate(c.tr_stream, off2_, std::min(part_n, n - off2_))) return -1;                                                     \
                    HIPCK(hipEventRecord(c.ev_part[p_ + 1], c.tr_stream));                                                                         \
                }                                                                                                                                  \
                HIPCK(hipStreamWaitEvent(st, c.ev_part[p_], 0));                                                                                   \
                if (p_) HIPCK(hipMemsetAsync(c.d_counters + C_ENCHUNK, 0, 4, st));                                                                 \
            }                                                                                                                                      \
            KERNEL<WV><<<dim3(blocks_), dim3(64 * WV), lds2, st>>>(h->d_T, X, h->d_bitmap, c.d_frames + off_ * 6 * FP, FP, L, cnt_, c.d_tasks, c.cap_tasks, c.d_counters, \
                                                                   c.d_stats, (uint32_t)off_, en_prio MC_FO_ARGS);                               \
            if (in_parts) HIPCK(hipEventRecord(c.ev_en[p_], st));                                                                                  \
        }                                                                                                                                          \
    } while (0)
        if (enq) { if (waves == 16) MC_LAUNCH_EN(k_enumerate_q, 16); else if (waves == 12) MC_LAUNCH_EN(k_enumerate_q, 12); else if (waves == 8) MC_LAUNCH_EN(k_enumerate_q, 8); else MC_LAUNCH_EN(k_enumerate_q, 4); }
        else { if (waves == 16) MC_LAUNCH_EN(k_enumerate_count, 16); else if (waves == 12) MC_LAUNCH_EN(k_enumerate_count, 12); else if (waves == 8) MC_LAUNCH_EN(k_enumerate_count, 8); else MC_LAUNCH_EN(k_enumerate_count, 4); }
#undef MC_LAUNCH_EN
    } else
        k_enumerate<<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st>>>(h->d_T, X, c.d_frames, FP, L, n, c.d_tasks, c.cap_tasks, c.d_counters, c.d_stats);
    HIPCK(hipEventRecord(c.ev[2], st));
    // the number of seed hits stays on the device: persistent workgroups walk the pool
    const size_t lds_ev = (size_t)(MC_EV_BS / 64) * MC_EV_QCAP * 32;   // a queue of survivors per wave: 32 KB per workgroup
    const bool ranges = h->fast_enum && !h->count_traffic;            // k_enumerate_q writes ranges of hits, the other two seed kernels single hits
    HIPCK(hipFuncSetAttribute(ranges ? (const void *)k_eval_seeds<true> : (const void *)k_eval_seeds<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_ev));
    static const unsigned ev_bpc = getenv("MC_EV_BPC") ? (unsigned)std::max(1, std::min(8, atoi(getenv("MC_EV_BPC")))) : (unsigned)MC_EV_BPC;   // (experiments)
    if (ranges) k_eval_seeds<true><<<dim3(256u * ev_bpc), dim3(MC_EV_BS), lds_ev, st>>>(h->d_T, X, c.d_frames, FP, L, c.d_tasks, c.d_counters + C_TASKS, c.cap_tasks, c.d_hsps, c.cap_hsps, c.d_gaps, c.cap_gaps, c.d_counters, h->d_P, h->d_fam, h->best_only ? c.d_cand : nullptr, c.d_hkeys, c.d_low, c.d_hplace);
    else k_eval_seeds<false><<<dim3(256u * ev_bpc), dim3(MC_EV_BS), lds_ev, st>>>(h->d_T, X, c.d_frames, FP, L, c.d_tasks, c.d_counters + C_TASKS, c.cap_tasks, c.d_hsps, c.cap_hsps, c.d_gaps, c.cap_gaps, c.d_counters, h->d_P, h->d_fam, h->best_only ? c.d_cand : nullptr, c.d_hkeys, c.d_low, c.d_hplace);
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
        // 2. order the flanks by DP size (a counting sort, k_gap_sort_*; the ordering kernels' buffers are idle at this point); 3. extend them with the DP rows
        // in LDS, those whose band leaves the window again with a wider one, the rest with full-size rows; 4. every task takes its
        // HSP from its group's flank results.  The counts of 2. - 4. stay on the device.
        static const int gap_refill = getenv("MC_GAP_REFILL") ? std::max(1, std::min(64, atoi(getenv("MC_GAP_REFILL")))) : MC_GAP_REFILL;   // (experiments)
        static const unsigned gap_wpc = getenv("MC_GAP_WPC") ? (unsigned)std::max(1, atoi(getenv("MC_GAP_WPC"))) : 8u;                      // waves per CU of the launch
        uint32_t slots = 1u << 16;
        while (slots < 2 * ngaps) slots <<= 1;
        if (slots > c.gtab_slots) { if (dalloc(&c.d_gtab, (size_t)slots)) return -1; c.gtab_slots = slots; }
        // The flank sort borrows the buffers of the HSP ordering: 2 ngaps keys in d_k64 (8 cap_hsps bytes), 2 ngaps items in d_idx /
        // d_idxo (4 cap_hsps bytes each).  The pools are sized so that ordinary batches fit (ensure_capacity); a batch
        // dense in gap tasks that does not is an overflow like any other: the range is run again in halves.
        if (2 * (uint64_t)ngaps > c.cap_hsps) { g_err = "gap task pool larger than the sort buffers"; return -2; }
        uint32_t *gk = (uint32_t *)c.d_k64, *gi = c.d_idx, *gio = c.d_idxo;
        HIPCK(hipMemsetAsync(c.d_gtab, 0, (size_t)slots * 8, st));
        HIPCK(hipMemsetAsync(c.d_ghist, 0, MC_GS_BINS * sizeof(uint32_t), st));
        k_gap_dedupe<<<dim3((ngaps + 255) / 256), dim3(256), 0, st>>>(X, L, c.d_gaps, ngaps, c.d_gtab, slots - 1, c.d_gleader, gk, gi, c.d_counters);
        k_gap_sort_hist<<<dim3(512), dim3(256), 0, st>>>(gk, c.d_counters + C_ITEMS, 2 * ngaps, c.d_ghist);
        k_gap_sort_scan<<<dim3(1), dim3(1024), 0, st>>>(c.d_ghist);
        k_gap_sort_scatter<<<dim3(512), dim3(256), 0, st>>>(gk, gi, c.d_counters + C_ITEMS, 2 * ngaps, c.d_ghist, gio);
        k_gapped_lds<MC_GAP_WIN, 64><<<dim3(std::min<uint32_t>((2 * ngaps + 63) / 64, 256u * gap_wpc)), dim3(64), 0, st>>>(h->d_T, X, c.d_frames, FP, L, c.d_gaps, gio, c.d_counters + C_ITEMS, c.d_fout,
                                                                                                                 c.d_counters + C_RETRY, c.d_retry, gap_refill, c.d_counters + C_GTAKE);
        k_gapped_lds<MC_GAP_WIN2, MC_GAP_LANES2><<<dim3(256u * 4u), dim3(64), 0, st>>>(h->d_T, X, c.d_frames, FP, L, c.d_gaps, c.d_retry, c.d_counters + C_RETRY, c.d_fout, c.d_counters + C_RETRY2, c.d_retry2, 1, c.d_counters + C_GTAKE2);
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
        hipStream_t side = order_serial ? st : c.side, side2 = order_serial ? st : h->best_only ? c.side : c.side2;   // (best hits only: few reads are ordered at all - a third stream only costs)
        if (!order_serial) { HIPCK(hipEventRecord(c.ev_fork, st)); HIPCK(hipStreamWaitEvent(c.side, c.ev_fork, 0)); HIPCK(hipStreamWaitEvent(c.side2, c.ev_fork, 0)); }
        HIPCK(hipFuncSetAttribute((const void *)k_order_heavy<1024, MC_ORDER_LDS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(MC_ORDER_LDS * 18)));
        k_order_heavy<1024, MC_ORDER_LDS><<<dim3(256u), dim3(1024), MC_ORDER_LDS * 18, side>>>(keys, c.d_places, slots, c.d_heads, heavy3, c.d_counters + C_ORDER3, c.d_counters + C_OTAKE3, c.d_low, order, c.d_gsz, c.d_nv, c.d_nrow, (uint64_t *)c.d_tmp);
        k_order_heavy<256, MC_ORDER_MID><<<dim3(256u * 4u), dim3(256), MC_ORDER_MID * 18, side2>>>(keys, c.d_places, slots, c.d_heads, heavy2, c.d_counters + C_ORDER2, c.d_counters + C_OTAKE2, c.d_low, order, c.d_gsz, c.d_nv, c.d_nrow, (uint64_t *)c.d_tmp);
        if (!order_serial) { HIPCK(hipEventRecord(c.ev_join, c.side)); HIPCK(hipEventRecord(c.ev_join2, c.side2)); }
        k_order_light<<<dim3((n + MC_OL_READS - 1) / MC_OL_READS), dim3(256), 0, st>>>(keys, c.d_places, slots, c.d_heads, n, c.d_low, order, c.d_gsz, c.d_nv, c.d_nrow);
        k_order_heavy<64, MC_ORDER_SMALL><<<dim3(256u * 16u), dim3(64), MC_ORDER_SMALL * 18, st>>>(keys, c.d_places, slots, c.d_heads, heavy, c.d_counters + C_ORDER, c.d_counters + C_OTAKE, c.d_low, order, c.d_gsz, c.d_nv, c.d_nrow, (uint64_t *)c.d_tmp);
        if (!order_serial) { HIPCK(hipStreamWaitEvent(st, c.ev_join, 0)); HIPCK(hipStreamWaitEvent(st, c.ev_join2, 0)); }
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
        uint32_t *d_heavy1 = c.d_retry2, *d_heap_order = c.d_retry2 + c.cap_gaps;   // (the ordering kernels' lists: done)
        const uint32_t light_pitch = (uint32_t)c.cap_reads + 1;
        k_heavy_lists<<<dim3((nheads + 255) / 256), dim3(256), 0, st>>>(c.d_nv, nheads, c.d_nrow, c.d_bestof, c.d_counters, d_heavy, d_light, light_pitch, h->best_only ? MC_FH_MIN_BEST : MC_FH_MIN,
                                                                         d_heavy1, d_heavy2, d_heavy3);
        HIPCK(hipEventRecord(c.ev_fork, st));
        {
            const size_t l1 = (size_t)MC_FH_N1 * 16 + 3 * (size_t)(MC_FH_N1 + 2) * 2, l2 = (size_t)MC_FH_N2 * 16 + 3 * (size_t)(MC_FH_N2 + 2) * 2, l3 = (size_t)MC_FH_N3 * 16 + 3 * (size_t)(MC_FH_N3 + 2) * 2;
            HIPCK(hipFuncSetAttribute((const void *)k_finish_heavy<MC_FH_N3, C_HEAVY3, -1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l3));
            HIPCK(hipStreamWaitEvent(c.side, c.ev_fork, 0));
            HIPCK(hipStreamWaitEvent(c.side2, c.ev_fork, 0));
            // the two kernels of the larger reads (few reads, long chains, a fraction of the GPU) beside the first one.  (Round 5, kernel trace:
            // second + third, 0.84 ms per 1 M reads, is the longer chain in front of MergeRes' heap sort; the third in front of the thread-per-read
            // kernels on this stream, or in front of the first on its stream, made the stage 0.1 - 0.2 ms LONGER - whatever runs behind the
            // third waits for its few long reads, and they hold 135 KB of a CU's LDS each.)
            const unsigned wpc2 = (unsigned)std::min<size_t>(8, std::max<size_t>(1, (size_t)(158 * 1024) / (l2 + 1024)));   // waves per CU the LDS holds
            // (the lists of the first two kernels with the longest stacks first - k_heavy_order; the third has a few dozen reads)
            static const bool fh_sort = !(getenv("MC_FH_SORT") && atoi(getenv("MC_FH_SORT")) == 0);
            uint32_t *d_sorted1 = c.d_retry2 + c.cap_gaps / 2, *d_sorted2 = c.d_retry2 + c.cap_gaps + c.cap_gaps / 2;      // (lists of at most n reads: cap_gaps >= 10 n)
            if (fh_sort) k_heavy_order<<<dim3(1), dim3(1024), 0, c.side2>>>(d_heavy2, c.d_counters + C_HEAVY2, d_heavy, c.d_nv, 2, d_sorted2);
            k_finish_heavy<MC_FH_N2, C_HEAVY2, -1><<<dim3(256 * wpc2), dim3(64), l2, c.side2>>>(h->d_T, X, h->d_P, h->d_fam, c.d_nv, c.d_heads, nheads, c.d_v, c.d_tmp, c.first_read_id,
                                                                                            c.d_nrow, c.d_bestof, c.d_counters, d_heavy, fh_sort ? d_sorted2 : d_heavy2, nullptr);
            k_finish_heavy<MC_FH_N3, C_HEAVY3, -1><<<dim3(256), dim3(64), l3, c.side2>>>(h->d_T, X, h->d_P, h->d_fam, c.d_nv, c.d_heads, nheads, c.d_v, c.d_tmp, c.first_read_id,
                                                                                         c.d_nrow, c.d_bestof, c.d_counters, d_heavy, d_heavy3, nullptr);
            HIPCK(hipEventRecord(c.ev_join2, c.side2));
            if (fh_sort) k_heavy_order<<<dim3(1), dim3(1024), 0, c.side>>>(d_heavy1, c.d_counters + C_HEAVY1, d_heavy, c.d_nv, 0, d_sorted1);
            k_finish_heavy<MC_FH_N1, C_HEAVY1, -1><<<dim3(256 * 12), dim3(64), l1, c.side>>>(h->d_T, X, h->d_P, h->d_fam, c.d_nv, c.d_heads, nheads, c.d_v, c.d_tmp, c.first_read_id,
                                                                                            c.d_nrow, c.d_bestof, c.d_counters, d_heavy, fh_sort ? d_sorted1 : d_heavy1, nullptr);
            HIPCK(hipStreamWaitEvent(c.side, c.ev_join2, 0));
            // MergeRes' heap sort of all of them (a lane per read), then their rows (a wave per read)
            const size_t lh = (size_t)(MC_MAX_M8 + 2) * 64 * 4;
            HIPCK(hipFuncSetAttribute((const void *)k_heap_lanes, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lh));
            k_heap_order<<<dim3(1), dim3(1024), 0, c.side>>>(d_heavy, c.d_nrow, c.d_counters, d_heap_order);
            k_heap_lanes<<<dim3(256), dim3(64), lh, c.side>>>(c.d_heads, nheads, nh, c.d_tmp, c.d_nrow, c.d_counters, d_heavy, d_heap_order);
            k_heavy_rows<<<dim3(256 * 12), dim3(64), 0, c.side>>>(h->d_T, X, h->d_P, h->d_fam, c.d_heads, nheads, c.d_v, c.d_tmp, c.first_read_id, c.d_nrow, c.d_bestof, c.d_counters, d_heavy);
            HIPCK(hipEventRecord(c.ev_join, c.side));
        }
        // the light reads: the four size classes side by side (the counts stay on the device; blocks past a class' count leave at once)
        {   // (size classes 2, 3 - up to 48 / MC_FH_MIN stacked HSPs - with MC_FH_MIN items of LDS per thread, classes 0, 1 - up to 4 / 16 - with 16; the
            // items are reached through generic pointers - mc_finish_stacked is shared with the host - and a flat access to LDS must stay
            // below 64 KB of the workgroup's allocation: 32 and 128 threads per workgroup)
            const size_t lb = 32 * (MC_FH_MIN * 16 + 16), ls = 128 * (16 * 16 + 16);
            static const int fin_lds = getenv("MC_FINISH_GLOBAL") ? 0 : 1;          // (experiments: the items in global scratch, as before round 4)
            HIPCK(hipFuncSetAttribute((const void *)k_finish<32, MC_FH_MIN, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb));
            k_finish<32, MC_FH_MIN, 2><<<dim3((nheads + 31) / 32, 2), dim3(32), lb, st>>>(h->d_T, X, h->d_P, h->d_fam, c.d_nv, c.d_heads, nheads, c.d_v, c.d_tmp,
                                                                                 c.first_read_id, c.d_nrow, c.d_bestof, d_light, light_pitch, c.d_counters + C_LIGHT0, fin_lds);
            k_finish<128, 16, 0><<<dim3((nheads + 127) / 128, 2), dim3(128), ls, st>>>(h->d_T, X, h->d_P, h->d_fam, c.d_nv, c.d_heads, nheads, c.d_v, c.d_tmp,
                                                                                     c.first_read_id, c.d_nrow, c.d_bestof, d_light, light_pitch, c.d_counters + C_LIGHT0, fin_lds);
        }
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
            const char *en[7] = {"loop, records asked", "survivors queued", "X-drop extension", "HSP, marks", "records written", "residue wait + seed", "growth, gate"};
            for (int k = 0; k < 7; k++) fprintf(stderr, "ev-timing %-21s total %9.1f Mcycles (lane 0 of every wave)\n", en[k], ev[k] / 1e6);
            unsigned long long z8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            HIPCK(hipMemcpyToSymbol(HIP_SYMBOL(g_ev_acc), z8, sizeof z8));
            unsigned long long tr[4];
            HIPCK(hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_ev_turns), sizeof tr));
            fprintf(stderr, "ev-turns forward: %.1f M lane-turns in %.2f M wave-turns = %.1f lanes of 64; backward: %.1f M in %.2f M = %.1f lanes\n", tr[0] / 1e6, tr[1] / 1e6, tr[1] ? (double)tr[0] / (double)tr[1] : 0.0,
                    tr[2] / 1e6, tr[3] / 1e6, tr[3] ? (double)tr[2] / (double)tr[3] : 0.0);
            HIPCK(hipMemcpyToSymbol(HIP_SYMBOL(g_ev_turns), z8, sizeof tr));
        }
        const char *nm[8] = {"group starts", "groups", "scan, items", "sort", "threshold, ranks", "heap sort", "rows", "other"};
        for (int k = 0; k < 8; k++) fprintf(stderr, "fh-timing %-17s total %9.1f Mcycles %9llu entries\n", nm[k], acc[k] / 1e6, cnt[k]);
        {
            unsigned long long w[12];
            HIPCK(hipMemcpyFromSymbol(w, HIP_SYMBOL(g_fh_worst), sizeof w));
            fprintf(stderr, "fh-worst read: %llu stacked HSPs, %.3f Mcycles:", w[0] & 0xFFFFF, (double)(w[0] >> 20) / 1e6);
            for (int k = 0; k < 8; k++) fprintf(stderr, " %s %.3f", nm[k], (double)w[1 + k] / 1e6);
            fprintf(stderr, "\n");
            unsigned long long z12[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            HIPCK(hipMemcpyToSymbol(HIP_SYMBOL(g_fh_worst), z12, sizeof z12));
        }
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        HIPCK(hipMemcpyToSymbol(HIP_SYMBOL(g_fh_acc), z, sizeof z)); HIPCK(hipMemcpyToSymbol(HIP_SYMBOL(g_fh_cnt), z, sizeof z));
    }
#endif
    HIPCK(hipMemcpyAsync(c.h_stats, c.d_stats, sizeof(unsigned long long) * S_N, hipMemcpyDeviceToHost, st));
    return counters_to_host(c);
}

// E: rows (final order and ABI layout: McRow == mc_row) and best hits into pinned host memory
static int stage_e(mc_handle *h, McCtx &c)
{
    hipStream_t st = c.stream;
    if (c.nrows) HIPCK(hipMemcpyAsync(h->pin_rows, c.d_rows, sizeof(McRow) * c.nrows, hipMemcpyDeviceToHost, h->rows_stream));   // (the range's stream has been waited for: d_rows is final)
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
static void best_materialize(mc_handle *h);

// A range whose seed hits / HSPs / rows overflow the pools sized for ordinary shotgun reads (-2 from the pipeline) is run again in
// halves, and their results joined: the caller sees one range either way.
extern "C" int mc_run_range(mc_handle *h, int64_t first, int64_t count, int64_t first_read_id)
{
    if (h && h->pipe_nout) { g_err = "mc_run_range: ranges begun with mc_range_begin() are still in flight"; return -1; }
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
        best_materialize(h);
        best.insert(best.end(), h->best.begin(), h->best.end());
        stats_add(tot, h->stats);
        off += nb;
    }
    h->res_rows = rows.data(); h->n_res_rows = (int64_t)rows.size(); h->best.swap(best); h->best_from = nullptr; h->stats = tot;
    return 0;
}

// A range in two halves.  range_begin ISSUES the front of the range (stage A: translation, seeds, seed evaluation - two thirds of
// its time) and returns at once; range_end does everything else: waits for A, issues and waits for B, C + D, sends the rows on their
// way and fetches the best hits.  mc_run_range is one after the other.  Callers with a stream of ranges (run_stream, bench.py) call
// end(i), begin(i + 1) and THEN look at the results of range i: the device works on the next front while the host collects rows
// and best hits (per 2 M reads of 150 bp 0.4 ms to bring the best hits into read order, whatever the caller does with them, and -
// mc_search with rows - the 270 MB of rows to copy out of the pinned buffer).  The next front may overwrite every pool of the
// range before: what the host still reads of it lies in pinned host memory (rows: two buffers in turn; best hits; the counters
// were taken at range_end), and the rows still leaving the device are waited for by stage D (ev_rows).
static int range_begin(mc_handle *h, McCtx &c, int64_t first, int64_t count, int64_t first_read_id)
{
    if (ensure_capacity(h, c, count)) return -1;
    c.reads = h->reads_dev + first * h->read_len; c.n = count; c.first_read_id = first_read_id;
    const int rc = stage_a(h, c);
    if (rc) { (void)hipStreamSynchronize(c.stream); return rc; }
    c.busy = true;
    return 0;
}

// the best hits of the range that ended last, in the order classify_reads meets the reads (input order) - made when somebody asks
static void best_materialize(mc_handle *h)
{
    McCtx *c = h->best_from;
    if (!c) return;
    h->best_from = nullptr;
    const uint32_t nb = h->best_count;                             // (taken at range_end: the front of the next range has reset the context's counts since)
    std::sort(c->h_best, c->h_best + nb, [](const McBestHit &x, const McBestHit &y) { return x.read < y.read; });
    h->best.resize(nb);
    for (uint32_t i = 0; i < nb; i++) { const McBestHit &x = c->h_best[i]; mc_best_hit &o = h->best[i]; o.read = x.read; o.family = x.family; o.aln = x.aln; o.target_len = x.target_len; o.bits = x.bits; }
}

static int range_end(mc_handle *h, McCtx &c)
{
    c.busy = false;
    memset(&h->stats, 0, sizeof h->stats);
    h->res_rows = nullptr; h->n_res_rows = 0; h->best.clear(); h->best_from = nullptr;
    h->stats.reads = c.n;
    int rc = stage_wait(c);
    if (rc == 0) rc = stage_b(h, c);
    if (rc == 0 && (rc = stage_wait(c)) == 0 && (rc = stage_c(h, c)) == 0) rc = stage_d(h, c);   // (C leaves its counts on the device: D is issued behind it)
    if (rc == 0) rc = stage_wait(c);
    if (rc == 0 && c.h_c[C_OVERFLOW]) { g_err = "row buffer overflow"; rc = -2; }
    if (rc) { (void)hipStreamSynchronize(c.stream); return rc; }
    h->pin_cur ^= 1; h->pin_rows = h->pin_slot[h->pin_cur]; h->pin_cap = h->pin_slot_cap[h->pin_cur];   // (the other slot may still be receiving the rows of the run before)
    c.nrows = (c.nh && !h->best_only) ? c.h_c[C_ROWS] : 0u; c.nsegs = c.h_c[C_SEGS]; c.nbest = c.h_c[C_BEST];
    if ((size_t)c.nrows > h->pin_cap) {                          // grow the pinned row buffer
        (void)hipStreamSynchronize(h->rows_stream);                  // (the copy of the run before writes into the other buffer: let it finish before anything is freed)
        const size_t want = (size_t)c.nrows + c.nrows / 4 + 1024;
        mc_row *nb = nullptr;
        if (hipHostMalloc((void **)&nb, want * sizeof(mc_row), hipHostMallocDefault) != hipSuccess) { g_err = "out of pinned host memory for the rows"; return -1; }
        if (h->pin_rows) (void)hipHostFree(h->pin_rows);
        h->pin_rows = nb; h->pin_cap = want; h->pin_slot[h->pin_cur] = nb; h->pin_slot_cap[h->pin_cur] = want;
        const int other = h->pin_cur ^ 1;                        // the other slot grows with it (pinning 300 MB takes 40 ms: not in the middle of a later run)
        if (h->pin_slot_cap[other] < want) {
            mc_row *ob = nullptr;
            if (hipHostMalloc((void **)&ob, want * sizeof(mc_row), hipHostMallocDefault) == hipSuccess) {
                if (h->pin_slot[other]) (void)hipHostFree(h->pin_slot[other]);
                h->pin_slot[other] = ob; h->pin_slot_cap[other] = want;
            }
        }
    }
    rc = stage_e(h, c);
    if (rc) { (void)hipStreamSynchronize(c.stream); (void)hipStreamSynchronize(h->rows_stream); return rc; }
    if (c.nrows) { HIPCK(hipEventRecord(h->ev_rows, h->rows_stream)); h->rows_pending = true; h->rows_ever = true; }
    if ((rc = stage_wait(c)) != 0) return rc;
    h->res_rows = h->pin_rows; h->n_res_rows = (int64_t)c.nrows;
    h->best_from = &c; h->best_count = c.nbest;                  // (mc_result_best_hits / whoever needs them: best_materialize)
#ifdef MC_EXP_TIMING
    { const char *nm[6] = {"staging/other", "append", "lookup", "push", "setup", "expand"}; for (int k = 0; k < 6; k++) fprintf(stderr, "timing %-14s %8.3f Mcycles/wave-avg  %10llu entries\n", nm[k], (double)c.h_stats[4 + k] / 4096.0 / 1e6, c.h_stats[10 + k]); }
#endif
    h->stats.bucket_lookups += (int64_t)c.h_stats[S_LOOKUPS]; h->stats.key_probes += (int64_t)c.h_stats[S_KEYPROBES]; h->stats.seed_tasks += (int64_t)c.h_stats[S_TASKS];
    h->stats.seed_exact_asks += (int64_t)c.h_stats[S_EXACT]; h->stats.seed_wild_asks += (int64_t)c.h_stats[S_WILD]; h->stats.seed_pair_asks += (int64_t)c.h_stats[S_PAIRS]; h->stats.seed_probes += (int64_t)c.h_stats[S_PROBES];
    h->stats.gap_tasks += c.ngaps - c.gpad; h->stats.hsps += c.nh_all; h->stats.rows += c.nrows; h->stats.reads_with_rows += c.nsegs;
    // kernel times: HIP events around the stages
    h->stats.ms_translate += ev_ms(c.ev[0], c.ev[1]); h->stats.ms_seed += ev_ms(c.ev[1], c.ev[2]); h->stats.ms_eval += ev_ms(c.ev[2], c.ev[3]);
    h->stats.ms_gapped += ev_ms(c.ev[3], c.ev[4]); h->stats.ms_sort += ev_ms(c.ev[4], c.ev[5]); h->stats.ms_finish += ev_ms(c.ev[5], c.ev[6]); h->stats.ms_total += ev_ms(c.ev[0], c.ev[6]);
    h->stats.classified = (int64_t)c.nbest;
    return 0;
}

static int range_check(mc_handle *h, int64_t first, int64_t count)
{
    if (!h || !h->run_set) { g_err = "mc_set_run() must be called first"; return -1; }
    if (first < 0 || count < 0 || first + count > h->nreads) { g_err = "range outside the resident read set"; return -1; }
    if (count > (1 << 21) - 1) { g_err = "batch larger than 2097151 reads"; return -1; }
    HIPCK(hipSetDevice(h->device));
    return 0;
}

static int run_range_once(mc_handle *h, int64_t first, int64_t count, int64_t first_read_id)
{
    if (range_check(h, first, count)) return -1;
    if (count == 0) { memset(&h->stats, 0, sizeof h->stats); h->res_rows = nullptr; h->n_res_rows = 0; h->best.clear(); h->best_from = nullptr; return 0; }
    McCtx &c = h->ctx[0];
    const int rc = range_begin(h, c, first, count, first_read_id);
    if (rc) return rc;
    return range_end(h, c);
}

// ---- a stream of ranges: the front of the next one issued before the host looks at the results of this one -------------------------
// mc_range_begin() enqueues the front of a range and returns at once; mc_range_end() completes it (results as after mc_run_range).
// end(i), begin(i + 1), results of i, end(i + 1), ... keeps the device busy while the host works on the results.  One range at a
// time is on the device: a second mc_range_begin() before mc_range_end() is refused.  A range that overflows a pool comes back with
// -2 from mc_range_end: it is no longer in flight; run it with mc_run_range (which splits it).
// (Round 4 also measured the TAIL of a range - ordering, finishing - running beside the front of the next, on ordinary streams, on
// streams of their own priority and on streams with CU masks: it does not pay)
extern "C" int mc_range_begin(mc_handle *h, int64_t first, int64_t count, int64_t first_read_id)
{
    if (range_check(h, first, count)) return -1;
    if (count <= 0) { g_err = "mc_range_begin: an empty range"; return -1; }
    if (h->pipe_nout) { g_err = "mc_range_begin: a range is in flight already (mc_range_end first)"; return -1; }
    const int rc = range_begin(h, h->ctx[0], first, count, first_read_id);
    if (rc) return rc;
    h->pipe_nout = 1;
    return 0;
}

extern "C" int mc_range_end(mc_handle *h)
{
    if (!h) { g_err = "null handle"; return -1; }
    if (h->pipe_nout == 0) { g_err = "mc_range_end: no range in flight"; return -1; }
    HIPCK(hipSetDevice(h->device));
    h->pipe_nout = 0;
    return range_end(h, h->ctx[0]);
}

extern "C" int mc_ranges_in_flight(const mc_handle *h) { return h ? h->pipe_nout : 0; }

// What the stages of the last mc_run_range() left on the device (the per-stage parity tests compare it with the CPU emulation of
// the same per-thread code - SURVEY.md 7.2): 0 the six frames of every read (rows of *record_bytes = FP bytes), 1 the seed kernel's
// hits (McSeedTask, 16 bytes; read = MC_TASK_NONE: padding of a block of the pool), 2 the gap tasks (McGapTask, 28 bytes), 3 the HSP
// pool (McHsp, 48 bytes: the ungapped HSPs of k_eval_seeds and those of the gapped stage).  Returns the bytes there are (copied if
// they fit cap_bytes), -1 on error.
extern "C" int64_t mc_debug_stage(mc_handle *h, int what, void *dst, int64_t cap_bytes, int32_t *record_bytes)
{
    if (!h || !h->run_set || what < 0 || what > 3) { g_err = "mc_debug_stage: bad argument"; return -1; }
    if (h->pipe_nout) { g_err = "mc_debug_stage: ranges are in flight"; return -1; }
    HIPCK(hipSetDevice(h->device));
    const McCtx &c = h->ctx[0];
    const void *src = nullptr; int64_t bytes = 0; int32_t rec = 0;
    if (what == 0) { src = c.d_frames; rec = h->FP; bytes = c.n * 6 * (int64_t)h->FP; }
    else if (what == 1) { src = c.d_tasks; rec = (int32_t)sizeof(McSeedTask); bytes = (int64_t)c.ntasks * rec; }
    else if (what == 2) { src = c.d_gaps; rec = (int32_t)sizeof(McGapTask); bytes = (int64_t)c.ngaps * rec; }
    else { src = c.d_hsps; rec = (int32_t)sizeof(McHsp); bytes = (int64_t)(c.nh_all + (c.h_c ? c.h_c[C_HPAD] : 0u)) * rec; }
    if (record_bytes) *record_bytes = rec;
    if (dst && bytes && bytes <= cap_bytes) { HIPCK(hipStreamSynchronize(c.stream)); HIPCK(hipMemcpy(dst, src, (size_t)bytes, hipMemcpyDeviceToHost)); }
    return bytes;
}

extern "C" int mc_set_counting(mc_handle *h, int on)
{
    if (!h) { g_err = "null handle"; return -1; }
    h->count_traffic = on != 0;
    return 0;
}

extern "C" int mc_run(mc_handle *h, int64_t first_read_id) { return h ? mc_run_range(h, 0, h->nreads, first_read_id) : -1; }

// The streaming form of the pipeline: batches of reads are fetched from a host-side source into pinned staging memory and
// uploaded by a thread of their own (two staging / device buffers in turn) while the calling thread runs the ranges of the batches
// before - upload and search overlap; and the front of batch k + 1 is issued before the results of batch k are collected
// (range_begin / range_end), so that the device does not wait for the host either.
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
    if (h->pipe_nout) { g_err = "ranges begun with mc_range_begin() are still in flight"; return -1; }
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
    MC_OT("run_stream: pools", t0);
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
    const uint8_t *saved_reads = h->reads_dev; const int64_t saved_n = h->nreads;
    // the results of the range that ended last -> those of the stream (called while the front of the next batch runs)
    bool pending = false;
    auto collect = [&]() {
        if (!pending) return;
        pending = false;
        if (h->keep_rows) { rows_wait(h); all_rows.insert(all_rows.end(), h->res_rows, h->res_rows + h->n_res_rows); }
        best_materialize(h);
        all_best.insert(all_best.end(), h->best.begin(), h->best.end());
        stats_add(tot, h->stats);
    };
    auto release = [&](int k) { std::unique_lock<std::mutex> lk(mu); slot[k].state = 0; cv.notify_all(); };
    // batch `k` is in flight: its range ends (a pool overflow is answered by mc_run_range: smaller ranges); its results are pending
    auto end_batch = [&](int k) -> int {
        int r = mc_range_end(h);
        if (r == -2) { h->reads_dev = slot[k].dev; h->nreads = slot[k].n; r = mc_run_range(h, 0, slot[k].n, first_read_id + slot[k].first); }
        if (r) return r;
        pending = true;
        release(k);                                                  // (the reads of the batch are no longer needed: the uploader may fill the slot)
        return 0;
    };
    int rc = 0, flying = -1;
    for (int k = 0;; k ^= 1) {
        {   // the next batch is not there yet (the source is the slower side): nothing to overlap with - finish the range in flight now
            bool ready;
            { std::unique_lock<std::mutex> lk(mu); ready = slot[k].state != 0; }
            if (!ready && flying >= 0) { if ((rc = end_batch(flying)) != 0) break; flying = -1; collect(); }
        }
        { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return slot[k].state != 0; }); }
        if (slot[k].state == 2) { if (slot[k].rc < 0) { rc = (int)slot[k].rc; if (!up_err.empty()) g_err = up_err; } break; }
        if (flying >= 0) { if ((rc = end_batch(flying)) != 0) break; flying = -1; }
        h->reads_dev = slot[k].dev; h->nreads = slot[k].n;
        if ((rc = mc_range_begin(h, 0, slot[k].n, first_read_id + slot[k].first)) != 0) break;
        flying = k;
        collect();                                                   // the batch before, while the front of this one runs
    }
    if (rc == 0 && flying >= 0) { rc = end_batch(flying); flying = -1; }
    if (rc == 0) collect();
    if (h->pipe_nout) (void)mc_range_end(h);                         // (after an error: nothing stays in flight)
    { std::unique_lock<std::mutex> lk(mu); abort_up = true; cv.notify_all(); }
    uploader.join();
    h->reads_dev = saved_reads; h->nreads = saved_n;
    if (rc) return rc;
    h->res_rows = all_rows.data(); h->n_res_rows = (int64_t)all_rows.size(); h->best.swap(all_best); h->best_from = nullptr; h->stats = tot;
    return 0;
}

void mc_host_copy(uint8_t *dst, const uint8_t *src, size_t bytes);   // (mc_reader.cpp: a batch copied by several threads)
extern "C" int mc_search(mc_handle *h, const uint8_t *reads, int64_t nreads, int64_t first_read_id)
{
    if (!h || !h->run_set) { g_err = "mc_set_run() must be called first"; return -1; }
    if (nreads < 0 || (nreads > 0 && !reads)) { g_err = "bad argument"; return -1; }
    const int64_t L = h->read_len;
    int64_t at = 0;
    return run_stream(h, [&](uint8_t *dst, int64_t max_reads, int64_t *first) -> int64_t {
        const int64_t n = std::max<int64_t>(0, std::min(max_reads, nreads - at));
        if (n > 0) mc_host_copy(dst, reads + at * L, (size_t)(n * L));
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
extern "C" int64_t mc_result_best_hits(mc_handle *h, const mc_best_hit **hits) { if (!h) return -1; best_materialize(h); *hits = h->best.data(); return (int64_t)h->best.size(); }
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
