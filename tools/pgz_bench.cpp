// Development aid (host only): throughput of the parallel inflate (csrc/mc_pgzip.h) on a .gz file, and the CRC-32 / length of what it
// delivers (compare with `gzip -dc file | cksum`-style references).   g++ -O3 -std=c++17 -o /tmp/pgz_bench tools/pgz_bench.cpp -lz -pthread
//   pgz_bench file.gz [threads [chunk_bytes [reps]]]        threads 0: the serial reader
#include "../microbecensus_amd/csrc/mc_pgzip.h"
#include <chrono>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    const int threads = argc > 2 ? atoi(argv[2]) : 8;
    const size_t chunk = argc > 3 ? (size_t)atol(argv[3]) : ((size_t)1 << 20);
    const int reps = argc > 4 ? atoi(argv[4]) : 3;
    int fd = open(argv[1], O_RDONLY);
    struct stat sb; fstat(fd, &sb);
    const uint8_t *m = (const uint8_t *)mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    std::vector<uint8_t> buf(4 << 20);
    for (int r = 0; r < reps; r++) {
        auto t0 = std::chrono::steady_clock::now();
        uint64_t total = 0; uint32_t crc = 0; bool bad = false; std::string msg;
        if (threads > 0) {
            mcgz::ParallelGz g(m, (size_t)sb.st_size, threads, chunk);
            if (!g.start()) { fprintf(stderr, "not taken\n"); return 1; }
            for (;;) { int n = g.read(buf.data(), (int)buf.size(), &bad, &msg); if (n > 0) { total += (uint64_t)n; if (r == 0) crc = (uint32_t)crc32(crc, buf.data(), (uInt)n); } if (n < (int)buf.size()) break; }
        } else {
            mcgz::SerialGz g(m, (size_t)sb.st_size);
            if (!g.start()) return 1;
            for (;;) { int n = g.read(buf.data(), (int)buf.size(), &bad, &msg); if (n > 0) { total += (uint64_t)n; if (r == 0) crc = (uint32_t)crc32(crc, buf.data(), (uInt)n); } if (n < (int)buf.size()) break; }
        }
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("threads %d: %llu bytes in %.3f s = %.0f MB/s%s%s", threads, (unsigned long long)total, dt, total / dt / 1e6, bad ? " BAD: " : "", bad ? msg.c_str() : "");
        if (r == 0) printf("  crc32 %08x", crc);
        printf("\n");
    }
    return 0;
}
