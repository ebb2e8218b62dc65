// Sanitizer driver for the host-side reader (TEST INFRASTRUCTURE; tests/test_sanitize.py builds it with mc_reader.cpp under
// -fsanitize=address,undefined and -fsanitize=thread - the GPU boxes run no sanitizers, the host code is where the threads are):
// the sampler with and without -d on a plain file, twice the file with the take inside, .gz (parallel inflate), .bz2 (block-parallel),
// describe / dupset walk / take over three windows, .bz2 block ranges, count_bases, quality offset.
// usage: reader_sanitize plain.fq file.fq.gz file.fq.bz2 read_len
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/mcensus.h"
static long run(const char *const *paths, int np, int L, long nreads, int fastq, int qoff, double minq, int dups)
{
    mc_reader *r = mc_reader_open(paths, np, L, nreads, fastq, qoff, minq, -5, 100, dups, nullptr);
    long n = mc_reader_run(r);
    mc_reader_stats st; mc_reader_get_stats(r, &st);
    unsigned long h = 1469598103934665603ul;
    const uint8_t *p = mc_reader_reads(r);
    for (long i = 0; i < n * L; i++) h = (h ^ p[i]) * 1099511628211ul;
    printf("  n=%ld records=%ld dups=%ld lowq=%ld hash=%lx err=%s\n", n, (long)st.records, (long)st.dups, (long)st.low_qual, h, n < 0 ? mc_reader_last_error() : "");
    mc_reader_close(r);
    return n;
}
int main(int argc, char **argv)
{
    const char *plain = argv[1], *gz = argv[2], *bz = argv[3];
    int L = atoi(argv[4]);
    const char *p1[1] = {plain}; const char *p2[1] = {gz}; const char *p3[1] = {bz}; const char *p4[2] = {plain, plain};
    printf("plain -d\n"); run(p1, 1, L, 1L << 40, 1, 33, 20, 1);
    printf("plain x2 -d take\n"); run(p4, 2, L, 100000, 1, 33, 20, 1);
    printf("plain\n"); run(p1, 1, L, 1L << 40, 1, 33, 20, 0);
    printf("gz -d\n"); run(p2, 1, L, 1L << 40, 1, 33, 20, 1);
    printf("bz2\n"); run(p3, 1, L, 1L << 40, 1, 33, 20, 0);
    printf("bz2 -d\n"); run(p3, 1, L, 1L << 40, 1, 33, 20, 1);
    // describe / walk / take over two windows
    {
        FILE *f = fopen(plain, "rb"); fseek(f, 0, SEEK_END); long size = ftell(f); fclose(f);
        mc_dupset *s = mc_dupset_open();
        long tot = 0;
        for (int w = 0; w < 3; w++) {
            mc_reader *r = mc_reader_open_range(plain, size * w / 3, size * (w + 1) / 3, L, 1L << 40, 1, 33, 20, -5, 100);
            const mc_rec_desc *d = nullptr;
            long n = mc_reader_describe(r, &d);
            std::vector<uint8_t> v(n > 0 ? n : 1);
            if (n > 0 && mc_dupset_walk(s, plain, d, n, v.data()) != 0) printf("walk failed %s\n", mc_reader_last_error());
            std::vector<uint8_t> out((size_t)(n > 0 ? n : 1) * L);
            long k = mc_reader_take(r, v.data(), n, 1L << 40, out.data());
            tot += k;
            mc_reader_close(r);
        }
        printf("describe/walk/take: %ld accepted\n", tot);
        mc_dupset_close(s);
    }
    {
        long nb = mc_bz2_blocks(bz);
        long tot = 0;
        for (int w = 0; w < 4 && nb > 0; w++) {
            long b0 = nb * w / 4, b1 = nb * (w + 1) / 4;
            if (b1 <= b0) continue;
            mc_reader *r = mc_reader_open_bz2_part(bz, b0, b1, '@', L, 1L << 40, 1, 33, 20, -5, 100);
            long n = mc_reader_run(r); tot += n > 0 ? n : 0;
            mc_reader_close(r);
        }
        printf("bz2 parts: blocks %ld accepted %ld\n", nb, tot);
    }
    {   // a .gz in three slices of chunks: all decode at once, the windows go along the chain, the CRCs at the end
        const long CB = 65536;
        long nc = mc_gz_chunks(gz, CB);
        mc_reader *r[3] = {nullptr, nullptr, nullptr};
        long tot = 0;
        int used = 0;
        for (int w = 0; w < 3 && nc >= 3; w++) {
            r[w] = mc_reader_open_gz_part(gz, nc * w / 3, nc * (w + 1) / 3, CB, '@', L, 1L << 40, 1, 33, 20, -5, 100);
            if (!r[w] || mc_reader_start(r[w]) != 0) { printf("gz part %d: %s\n", w, mc_reader_last_error()); break; }
            used++;
        }
        std::vector<uint8_t> st(32768 + 64);
        for (int w = 0; w + 1 < used; w++) {
            long n = mc_reader_gz_end_state(r[w], st.data(), (long)st.size());
            mc_reader_gz_provide(r[w + 1], st.data(), n > 0 ? n : 0);
        }
        uint8_t crc[12] = {0}, crc2[12];
        for (int w = 0; w < used; w++) {
            long n = mc_reader_join(r[w]); tot += n > 0 ? n : 0;
            if (mc_reader_gz_finish(r[w], crc, crc2) != 0) printf("gz finish %d: %s\n", w, mc_reader_last_error());
            for (int i = 0; i < 12; i++) crc[i] = crc2[i];
        }
        for (int w = 0; w < 3; w++) if (r[w]) mc_reader_close(r[w]);
        printf("gz parts: chunks %ld accepted %ld\n", nc, tot);
    }
    printf("count_bases %ld qoff %d\n", (long)mc_count_bases(p3, 1), mc_quality_offset(gz));
    return 0;
}
