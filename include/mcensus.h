/* mcensus.h - C ABI of libmcensus_hip.so: the MI355X (gfx950) implementation of MicrobeCensus' hot path.
 *
 * The reference has no FFI: its hot path is a subprocess,
 *     rapsearch -q tmp -d rapdb_2.15 -o tmp -z T -e 1 -t n -p f -b 0
 * launched by search_seqs()            (/root/reference/microbe_census/microbe_census.py:369-389),
 * whose tmp.m8 is parsed by parse_rapsearch()/classify_reads()          (microbe_census.py:391-460),
 * and whose database comes from `prerapsearch -d markers.faa -n rapdb_2.15` (training/search_reads.py:57
 * documents the flag set).  Each entry point below names the piece of that interface it replaces.
 * Plain pointers and sizes only; every call returns 0 (or a count) on success and a negative value on
 * error, with the message available from mc_last_error().  There is no CPU fallback: without a HIP device
 * mc_open() fails.
 */
#ifndef MCENSUS_H
#define MCENSUS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct mc_handle mc_handle;

/* One row of RAPsearch2's m8 output (the 12 columns parse_rapsearch reads, microbe_census.py:391-398),
 * kept binary: query = read id, subject = marker index. */
typedef struct mc_row {
    int32_t query, subject;
    double ident;
    int32_t alnlen, mismatch, gapopen, qstart, qend, sstart, send;
    double loge, bits;
    int32_t score, nmatch;
} mc_row;

/* best_hits[read] = [family, aln, aln/target_len, score]   (classify_reads, microbe_census.py:450-453) */
typedef struct mc_best_hit {
    int32_t read, family, aln, target_len;
    double bits;
} mc_best_hit;

typedef struct mc_stats {
    int64_t reads, seed_tasks, gap_tasks, hsps, rows, reads_with_rows, classified;
    int64_t bucket_lookups, key_probes;   /* algorithmic traffic of the seed kernel: 8 B and 2 B reads */
    float ms_translate, ms_seed, ms_eval, ms_gapped, ms_sort, ms_finish, ms_total;
    /* what the seed kernel itself asked its structures (timed form): 9-mer filter words (4 B), wildcard filter lines (32 B),
     * pair filter blocks (16 B), bucket records + key groups searched (32 B + 16 B); seed_tasks postings (4 B) came out */
    int64_t seed_exact_asks, seed_wild_asks, seed_pair_asks, seed_probes;
    /* how often a range overflowed the pools sized for shotgun reads and was run again in smaller pieces (mc_run_range) */
    int64_t range_splits;
} mc_stats;

const char *mc_last_error(void);
int mc_device_count(void);

/* Replaces `prerapsearch -d <fasta> -n <db>` + the DB load of rapsearch (CHashSearch::BuildDHash / Search):
 * builds the reduced-alphabet 6-mer index of the marker proteins on the host and uploads it to `device`.
 * marker_family[i] is the gene family index of marker i (gene_fam.map, microbe_census.py:438). */
mc_handle *mc_open(const char *const *names, const char *const *seqs, int32_t nseq,
                   const int32_t *marker_family, int32_t nfam, int32_t device);
void mc_close(mc_handle *h);
/* mc_open() spends half a second of host time building the index (buckets, suffix keys, filters) - more than the search of the
 * reference's default run of 1 - 2 M reads (scripts/run_microbe_census.py:31: one run_pipeline per process).  With a cache directory
 * set (process-wide; NULL or "": none) the built index is kept there in a file named by a hash of the marker names and sequences
 * and read back by later mc_open() calls; a file that does not match its header, sizes, input hash and checksum is ignored and
 * rebuilt.  The directory should be writable by the user alone (the Python layer uses ~/.cache/microbecensus_amd, mode 0700). */
int mc_set_index_cache(const char *dir);
/* Host only (tests; no GPU): builds the index, writes it to dir, reads it back and compares every array, then checks that a file of
 * other sequences, a damaged and a truncated file are refused.  0 = all of that held. */
int mc_index_cache_check(const char *const *names, const char *const *seqs, int32_t nseq, const char *dir);

/* The same from a database `prerapsearch` already wrote (the reference ships one as data/rapdb_2.15): residues, buckets,
 * posting order and suffix keys are taken from the file, the GPU-side structures derived from them.  All markers start in
 * family 0 of 1; name them with mc_marker_name() and assign the families with mc_set_families() (gene_fam.map), then
 * mc_set_run().  mc_rapdb_verify() needs no GPU: 0 if the file holds exactly the index mc_open() builds from the sequences. */
mc_handle *mc_open_rapdb(const char *rapdb_path, int32_t device);
int32_t mc_marker_count(const mc_handle *h);
const char *mc_marker_name(const mc_handle *h, int32_t i);
int mc_set_families(mc_handle *h, const int32_t *marker_family, int32_t nfam);
int mc_rapdb_verify(const char *rapdb_path, const char *const *names, const char *const *seqs, int32_t nseq);
/* ... and the writer (no GPU): what `prerapsearch -d <fasta> -n <path>` produces, <path> and <path>.info. */
int mc_rapdb_write(const char *const *names, const char *const *seqs, int32_t nseq, const char *path);

/* Host views of the index, for cross-checking against a prerapsearch-built database (tests only). */
int mc_index_view(const mc_handle *h, const uint8_t **res_codes, const uint32_t **offsets, const uint32_t **bucket_starts,
                  const uint32_t **postings, const uint16_t **keys, int64_t *nres, int64_t *npostings,
                  uint32_t *freq_thr, double letter_p[10]);

/* Per-run parameters: trimmed read length (args['read_length']), the -e threshold, and
 * find_opt_pars(pars.map, L) (microbe_census.py:61-72) as arrays indexed by family. aln_stat: 0 hits, 1 cov, 2 aln. */
int mc_set_run(mc_handle *h, int32_t read_len, double loge_thr, const double *min_cov, const double *min_score,
               const int32_t *max_aaid, const int32_t *aln_stat);

/* Replaces search_seqs() + classify_reads(): reads = nreads x read_len bytes (the trimmed sequences that
 * process_seqfile writes, one after another, no separators).  Runs the whole device pipeline.
 * first_read_id is added to the read index to form the query id. */
int mc_search(mc_handle *h, const uint8_t *reads, int64_t nreads, int64_t first_read_id);

/* Same pipeline on reads that are already resident in HBM (bench / streaming):
 * mc_upload() copies a batch to the device, mc_run() executes the kernels on it (no host transfers of reads). */
int mc_upload(mc_handle *h, const uint8_t *reads, int64_t nreads);
/* ... or adopts caller-owned device memory (e.g. a torch uint8 tensor) as the resident read set. */
int mc_attach(mc_handle *h, const void *device_reads, int64_t nreads);
int mc_run(mc_handle *h, int64_t first_read_id);
/* runs the pipeline on reads [first, first+count) of the resident set (count <= 2097151). */
int mc_run_range(mc_handle *h, int64_t first, int64_t count, int64_t first_read_id);

/* A stream of ranges without the device waiting for the host.  mc_range_begin() enqueues the FRONT of a range (translation,
 * seeds, seed evaluation: two thirds of its time) and returns at once; mc_range_end() completes it - results as after
 * mc_run_range(), valid until the next mc_range_end() / mc_run_range().  The order end(i), begin(i + 1), <look at the results
 * of i>, end(i + 1), ... lets the host collect rows and best hits while the device works on the next front (what the host still
 * reads of range i lies in pinned host memory; the pools on the device are free for range i + 1).  One range at a time is on the
 * device: a second mc_range_begin() before mc_range_end() is refused.  mc_search() / mc_search_files() run their batches this
 * way, bench.py its steps.  mc_range_end() returns -2 when the range overflowed a pool: it is then no longer in flight - give
 * it to mc_run_range(), which runs it in smaller pieces.  mc_run_range() and mc_search*() refuse to run while a range is in
 * flight.  Results never depend on any of this. */
int mc_range_begin(mc_handle *h, int64_t first, int64_t count, int64_t first_read_id);
int mc_range_end(mc_handle *h);
int mc_ranges_in_flight(const mc_handle *h);

/* Test aid (tests/test_gpu_parity.py, per-stage parity with the CPU emulation of the kernels' per-thread code): what the stages of
 * the last mc_run_range() left on the device - what = 0: the six translated, SEG-masked frames of every read (rows of
 * *record_bytes bytes; BuildQHash@0x40b530, Seg::*), 1: the seed hits (16-byte records; Searching@0x415050), 2: the gap tasks
 * (28 bytes; ExtendSeq2Set@0x413b90), 3: the HSP pool (48 bytes; CalRes@0x4077a0).  Returns the number of bytes (copied to dst
 * when they fit cap_bytes; dst may be NULL to ask for the size), -1 on error. */
int64_t mc_debug_stage(mc_handle *h, int what, void *dst, int64_t cap_bytes, int32_t *record_bytes);

/* The seed kernel can count the index reads of the reference's algorithm for the batch (mc_stats.bucket_lookups /
 * key_probes: what CHashSearch::Searching@0x415050 / ExtendSeq2Set@0x413b90 would read - bench.py reports the rate at which
 * the timed kernel disposes of them).  Off by default: the two fields stay 0 and the kernel answers most one-substitution
 * probes from its wildcard and pair filters instead of searching them (mc_stats.seed_* count what it asks).  With on != 0
 * every probe is searched and counted.  Results do not depend on it. */
int mc_set_counting(mc_handle *h, int on);

/* Results of the last mc_search()/mc_run(), owned by the handle until the next call:
 * rows in the reference's m8 order (ascending read id, then RAPsearch2's order within a read); best hits in
 * ascending read id (only reads with a passing hit). */
int64_t mc_result_rows(mc_handle *h, const mc_row **rows);
int64_t mc_result_best_hits(mc_handle *h, const mc_best_hit **hits);
int mc_result_stats(mc_handle *h, mc_stats *out);

/* Writes the rows of the last run as RAPsearch2 m8 text (PrintRes formatting: %g columns, tab separated,
 * no header lines) - what search_seqs() leaves in paths['tempfile']+'.m8'. append != 0 appends. */
int mc_write_m8(mc_handle *h, const char *path, int append);
/* The same with the Query column taken from query_names[query id - first_read_id] (rapsearch prints the FASTA header's first
 * token; process_seqfile names its reads 0, 1, ... so the two agree there) - for the rapsearch-compatible executable. */
int mc_write_m8_named(mc_handle *h, const char *path, int append, const char *const *query_names, int64_t n_names, int64_t first_read_id);

/* The training workflow's grid search (training/training.py:311-334 classify_reads, called by training/class_reads.py:51-66 with
 * 4 aln_covs x 6 max_pids x 27 min_scores) over the m8 rows of the last mc_search() / mc_run(): for every combination, the rows
 * that pass alignment coverage >= aln_cov, identity <= max_pid (integers, as in class_reads.py), bit score >= min_score; per
 * read the best-scoring survivor (the first on a tie); per family the number of such reads, the sum of their alignment lengths,
 * the sum of alignment length / target length.  Outputs are [n_cov][n_pid][n_score][nfam] arrays; hits and aln are exact, the
 * coverage sums are accumulated in no fixed order (1e-12 relative against the reference's sequential sum).  One device pass. */
int mc_grid_classify(mc_handle *h, const double *aln_covs, int32_t n_cov, const int32_t *max_pids, int32_t n_pid, const double *min_scores, int32_t n_score,
                     int64_t *count_hits, int64_t *count_aln, double *count_cov);

/* ---- host stage in front of the search: native read sampler (csrc/mc_reader.cpp; no GPU involved) ----------------
 * Replaces open_file / parse_seqs / quality_filter / process_seqfile (microbe_census.py:47-59, :294-325, :265-279,
 * :328-367) and count_bases (:573-584) with identical results, quirks included (see the header of mc_reader.cpp).
 * Plain, .gz and .bz2 inputs (libbz2 is bound at run time).  Errors: NULL / negative + mc_reader_last_error(); -3 = the
 * reference would have raised inside run_pipeline (its message names the Python exception). */
typedef struct mc_reader mc_reader;
typedef struct mc_reader_stats {
    int64_t sampled, too_short, low_qual, dups;   /* the four counts process_seqfile prints (:362-366) */
    int64_t records;                              /* records parsed before the sampler stopped */
    int64_t bases;                                /* their total sequence length ... */
    int64_t exhausted;                            /* ... which is count_bases() (:573-584) when every file was read to its end (1) */
    int64_t ragged_end;                           /* the data ended inside a FASTQ record (a truncated file; a byte window that does not end on a record boundary) */
} mc_reader_stats;

const char *mc_reader_last_error(void);
/* Caps the worker threads of the host stages (record parsing, parallel inflate) of every reader opened afterwards: the reference's
 * args['threads'] (-t, microbe_census.py:270, forwarded there to rapsearch -z).  n <= 0: the CPUs the process may use (cgroup quota), up to 32 (default). */
void mc_set_host_threads(int32_t n);
/* fastq: args['file_type'] == 'fastq'; quality_offset: 32 or 64 as auto_detect_quality_offset returns it (:175-187);
 * fasta_out (may be NULL): the temp FASTA process_seqfile writes, ">{id}\n{seq[:L]}\n" per accepted read. */
mc_reader *mc_reader_open(const char *const *paths, int32_t npaths, int32_t read_len, int64_t nreads, int32_t fastq, int32_t quality_offset,
                          double min_quality, double mean_quality, double max_unknown, int32_t filter_dups, const char *fasta_out);
/* The same sampler on the byte window [byte_lo, byte_hi) of ONE plain (uncompressed, regular) file: the records that start in it.
 * Both ends are moved to the first record start behind them by one rule ('@' line whose second next line starts with '+' in a file
 * that starts with '@'; '>' line in one that starts with '>'), so consecutive windows cut a file into whole records whoever reads
 * them: the ranks of a multi-GPU run sample their own slices side by side (microbecensus_amd/distributed.py) and reproduce the
 * head-take of process_seqfile (:337-356) from the per-slice counts.  No duplicate filter (it needs the whole stream in one place). */
mc_reader *mc_reader_open_range(const char *path, int64_t byte_lo, int64_t byte_hi, int32_t read_len, int64_t nreads, int32_t fastq,
                                int32_t quality_offset, double min_quality, double mean_quality, double max_unknown);
/* -d (filter_dups) with a sampler on every rank.  The duplicate rule of process_seqfile (:345 the test comes before the quality filter,
 * :354 only accepted reads enter the set) is class-local: a record's fate depends on nothing but the earlier records with the same
 * sequence or its reverse complement.  So parsing, quality filter and hashing run on every rank's own byte window
 * (mc_reader_open_range + mc_reader_describe: one 32-byte descriptor per record), the ranks exchange the descriptors, every rank gives
 * every record of the round its verdict with mc_dupset_walk (same input, same verdicts; sequences are compared on the file's own
 * mapping) and copies the accepted reads of its own window with mc_reader_take (microbecensus_amd/distributed.py).
 * flags / verdict bits: 1 too short, 2 has qualities, 4 a base outside ACGTN, 8 accepted, 16 fails the quality filter, 32 duplicate,
 * 64 the reference raises at this record. */
typedef struct mc_rec_desc { uint64_t h1, h2; uint64_t seq_off; uint32_t len; uint8_t flags, pad[3]; } mc_rec_desc;
typedef struct mc_dupset mc_dupset;
int64_t mc_reader_describe(mc_reader *r, const mc_rec_desc **out);   /* records of the window (array owned by the reader), or < 0; mc_reader_stats.ragged_end: not usable, fall back */
mc_dupset *mc_dupset_open(void);
void mc_dupset_close(mc_dupset *s);
int mc_dupset_walk(mc_dupset *s, const char *path, const mc_rec_desc *d, int64_t n, uint8_t *verdict);   /* n descriptors of `path` in file order -> n verdicts; the set remembers across calls and files */
int64_t mc_reader_take(mc_reader *r, const uint8_t *verdict, int64_t n, int64_t max_take, uint8_t *dst);   /* accepted reads of the described window (first read_len bases), at most max_take */
/* The same sampler on the blocks [block_lo, block_hi) of ONE .bz2 file (open_file :55-58): the records that start in the TEXT of those
 * blocks, both ends moved to the first record start behind them by the rule of mc_reader_open_range.  The blocks of a bzip2 file are
 * independent (csrc/mc_pbzip2.h), so the ranks of a multi-GPU run decode and sample their own shares side by side - which a .gz does not
 * allow.  kind: '@' or '>' (what a record of the file starts with).  mc_bz2_blocks(): the number of blocks of a file all of whose streams
 * check out from its first to its last byte, or -1 (cut short, damaged, trailing bytes: one sampler reads such a file and reports what the
 * reference would).  No duplicate filter. */
mc_reader *mc_reader_open_bz2_part(const char *path, int64_t block_lo, int64_t block_hi, int32_t kind, int32_t read_len, int64_t nreads, int32_t fastq,
                                   int32_t quality_offset, double min_quality, double mean_quality, double max_unknown);
int64_t mc_bz2_blocks(const char *path);
/* A .gz file across the ranks.  A gzip member cannot be entered in the middle (every block may point 32 KB back) but it can be DECODED
 * from the middle speculatively (csrc/mc_pgzip.h): the file is cut into slices of chunks, every rank decodes its slice at once, and what
 * is sequential is a chain of hand-overs - where the slice in front ended and the 32 KB in front of that - after which every rank samples
 * the records that start in its slice's text (both ends moved to the first record start behind them, as mc_reader_open_range).
 *   mc_gz_chunks            chunks of chunk_bytes the parallel reader cuts the file into, or -1 (not a gzip file it takes)
 *   mc_reader_open_gz_part  the sampler on chunks [chunk_lo, chunk_hi); run it with mc_reader_start / mc_reader_join (it waits for the state)
 *   mc_reader_gz_provide    what mc_reader_gz_end_state of the slice in front returned (n = 0: that slice failed); not needed for chunk_lo = 0
 *   mc_reader_gz_end_state  waits until the slice is stitched; returns the bytes written (at most 32784), or -1
 *   mc_reader_gz_finish     after mc_reader_join: checks the CRCs of the members that end in the slice, given CRC (4 bytes) | length (8 bytes) of
 *                           the open member's bytes in front of it (zeros for the first slice); 0, or -3 = gzip.open's "CRC check failed" */
int64_t mc_gz_chunks(const char *path, int64_t chunk_bytes);
mc_reader *mc_reader_open_gz_part(const char *path, int64_t chunk_lo, int64_t chunk_hi, int64_t chunk_bytes, int32_t kind, int32_t read_len, int64_t nreads,
                                  int32_t fastq, int32_t quality_offset, double min_quality, double mean_quality, double max_unknown);
int mc_reader_gz_provide(mc_reader *r, const uint8_t *state, int64_t n);
int64_t mc_reader_gz_end_state(mc_reader *r, uint8_t *out, int64_t cap);
int mc_reader_gz_finish(mc_reader *r, const uint8_t *crc_in, uint8_t *crc_out);
/* Runs the sampler: returns args['sampled_reads'] (0 = "No reads remaining after filtering"). */
int64_t mc_reader_run(mc_reader *r);
/* sampled x read_len bytes, row i = trimmed read i: exactly what mc_search() / mc_upload() take. Owned by the reader. */
const uint8_t *mc_reader_reads(mc_reader *r);
int mc_reader_get_stats(mc_reader *r, mc_reader_stats *out);
/* Seconds of the last mc_reader_run() by phase, up to n values (returns how many were written): [0] the whole run; on the sampler's own
 * thread: [1] waiting for input (inflate), [2] guessing the pieces' starts, [3] the parse (all workers), [4] stitching the pieces,
 * [5] verdicts, places and copies of the accepted reads, [6] of [5]: the walkers of the duplicate classes (-d; process_seqfile :345, :354). */
int32_t mc_reader_times(const mc_reader *r, double *out, int32_t n);
void mc_reader_close(mc_reader *r);
/* A closed reader leaves its read buffer (touched pages) to the next reader of the process: the second run_pipeline() of a process pays
 * neither munmap nor page faults.  At most keep_bytes of it are kept (default 4 GB); mc_reader_trim() sets the limit and releases
 * what is held beyond it (0: everything). */
void mc_reader_trim(int64_t keep_bytes);
/* count_bases(): total sequence length over every record of every file. */
int64_t mc_count_bases(const char *const *paths, int32_t npaths);
/* auto_detect_quality_offset() (microbe_census.py:175-187): 32 or 64 by the first quality character of the file that decides
 * (32 when none does); -2 = a record without a quality line (take the Python path: it fails the way the reference does). */
int32_t mc_quality_offset(const char *path);

/* Streaming form of the sampler: mc_reader_start() runs it on a thread of its own; mc_reader_fetch() blocks until reads
 * [first, first + max_reads) are sampled (or the sampler has ended), copies them to dst and returns how many there were
 * (0: no more; negative: the sampler's error); mc_reader_join() waits for the end and returns what mc_reader_run() would have. */
int mc_reader_start(mc_reader *r);
int64_t mc_reader_fetch(mc_reader *r, int64_t first, int64_t max_reads, uint8_t *dst);
int64_t mc_reader_join(mc_reader *r);
int32_t mc_reader_read_len(const mc_reader *r);
/* the nreads the reader was opened with: how many reads it delivers at most (mc_search_files sizes its buffers by it) */
int64_t mc_reader_nreads(const mc_reader *r);

/* process_seqfile() + search_seqs() + classify_reads() in one call (microbe_census.py:328-460): the sampler runs beside the
 * search, batches of accepted reads go through pinned staging memory to the GPU while the batch before is searched.  Results as
 * after mc_search(); the reader's statistics (mc_reader_get_stats) are complete when it returns.  -3: the reference would have
 * raised while sampling (mc_last_error names the Python exception). */
int mc_search_files(mc_handle *h, mc_reader *r, int64_t first_read_id);
/* The same over n_dev GPUs of this process (the multi-device entry; the one-process-per-GPU form with an RCCL
 * reduce is microbecensus_amd/distributed.py): handles[d] was opened on device d and given the same mc_set_run(); the sampler
 * runs once, batches of accepted reads are dealt to the devices as they ask for them (global read ids), one host thread per
 * device.  Results stay with the handles (mc_result_* per handle): the caller adds the per-family sums - integers. */
int mc_search_files_multi(mc_handle *const *handles, int32_t n_dev, mc_reader *r, int64_t first_read_id);
/* keep != 0 (default): mc_search() / mc_search_files() collect the m8 rows of all their batches for mc_result_rows(); 0: only the
 * best hits and the statistics (the rows are still computed - classification reads them on the device). */
int mc_set_keep_rows(mc_handle *h, int keep);
/* on != 0: the runs that follow produce the best hits only (what classify_reads :432-460 keeps) - no m8 rows.  A read none of
 * whose HSPs would pass its family's thresholds (min_cov, max_aaid, min_score) as an m8 row cannot be classified, whatever its
 * ranking: only the reads that have such an HSP are sorted and finished (with ALL their HSPs: ranking, sum statistics and the
 * 500-row cap are the reference's).  mc_result_best_hits() is identical to the full path's; mc_result_rows() is empty and
 * mc_stats.rows / reads_with_rows count the finished reads only.  run_pipeline() uses it when it is not asked for the
 * "reads hit marker proteins" line (verbose). */
int mc_set_best_hits_only(mc_handle *h, int on);

#ifdef __cplusplus
}
#endif
#endif
