// tools/gather_ceiling.hip - how many SCATTERED cache lines per second one MI355X delivers, by footprint and item width: the
// ceiling the seed kernel's asks are priced against (development aid, run on the GPU box by tools/gather_ceiling.sh;
// not part of the product).
//   k_gather<BYTES>   every lane asks one aligned item of BYTES at a place that is a hash of its request number: nothing depends
//                     on what comes back, UNROLL asks of a lane are in flight together (the memory system's rate, not its latency)
//   k_chase           every lane's next place is a hash of what the last one returned (one ask in flight per lane: the latency of
//                     the level the footprint lives in, and through Little's law the rate a dependent cascade of W waves can reach)
//   k_mix             per turn and lane the seed kernel's mix for a 150 bp read (per 11 asks: 4 words out of 128 KB - bucket
//                     bitmap -, 2 words out of 1 MB - 9-mer filter -, 2 lines of 32 B out of 16 MB - wildcard filter -, 1 block of
//                     16 B out of 16 MB - pair filter -, 2 words out of 80 MB - bucket records, keys, postings, offsets)
// Durations with HIP events around REP launches; the grid is WPC waves per CU on 256 CUs, persistent.  One JSON document on stdout.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
template <int BYTES> __device__ __forceinline__ uint32_t ask(const uint8_t *__restrict__ p, size_t at)
{
    if (BYTES == 32) { const uint4 a = *(const uint4 *)(p + at), b = *(const uint4 *)(p + at + 16); return a.x ^ a.w ^ b.x ^ b.w; }
    if (BYTES == 16) { const uint4 a = *(const uint4 *)(p + at); return a.x ^ a.w; }
    if (BYTES == 8) { const uint2 a = *(const uint2 *)(p + at); return a.x ^ a.y; }
    return *(const uint32_t *)(p + at);
}
template <int BYTES, int UNROLL>
__global__ void k_gather(const uint8_t *__restrict__ p, uint32_t item_mask, uint32_t per_lane, uint32_t *out)
{
    uint32_t acc = 0;
    const uint64_t lane = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t i = 0; i < per_lane; i += UNROLL) {
        uint32_t v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = ask<BYTES>(p, (size_t)((uint32_t)mix((lane << 20) + i + u) & item_mask) * BYTES);
#pragma unroll
        for (int u = 0; u < UNROLL; u++) acc += v[u];
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void k_chase(const uint8_t *__restrict__ p, uint32_t item_mask, uint32_t per_lane, uint32_t *out)
{
    const uint64_t lane = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t at = (uint32_t)mix(lane) & item_mask, acc = 0;
    for (uint32_t i = 0; i < per_lane; i++) {
        const uint32_t v = *(const uint32_t *)(p + (size_t)at * 4);              // (the buffer holds the byte 1: v is a constant the compiler cannot know)
        acc += v;
        at = (uint32_t)mix((lane << 20) + i + v) & item_mask;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
// DEP = 0: the eleven asks of a turn are independent; DEP = 1: they form the cascade's chain of four levels (bitmap + 9-mer
// filter -> wildcard lines -> pair block -> records), each level's places a hash of the level before's answers
template <int DEP>
__global__ void k_mix(const uint8_t *__restrict__ A, const uint8_t *__restrict__ B, const uint8_t *__restrict__ C, const uint8_t *__restrict__ D,
                      const uint8_t *__restrict__ E, uint32_t e_mask, uint32_t per_lane, uint32_t *out)
{
    const uint64_t lane = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (uint32_t i = 0; i < per_lane; i++) {
        const uint64_t h0 = mix((lane << 24) + (uint64_t)i * 4), h1 = mix(h0 + 1);
        uint32_t a = ask<4>(A, (size_t)((uint32_t)h0 & 0x7FFFu) * 4) + ask<4>(A, (size_t)((uint32_t)(h0 >> 16) & 0x7FFFu) * 4)
                   + ask<4>(A, (size_t)((uint32_t)(h0 >> 32) & 0x7FFFu) * 4) + ask<4>(A, (size_t)((uint32_t)(h0 >> 48) & 0x7FFFu) * 4);
        a += ask<4>(B, (size_t)((uint32_t)h1 & 0x3FFFFu) * 4) + ask<4>(B, (size_t)((uint32_t)(h1 >> 32) & 0x3FFFFu) * 4);
        const uint64_t h2 = mix(h1 + (DEP ? a : 0u));
        uint32_t c = ask<32>(C, (size_t)((uint32_t)h2 & 0x7FFFFu) * 32) + ask<32>(C, (size_t)((uint32_t)(h2 >> 32) & 0x7FFFFu) * 32);
        const uint64_t h3 = mix(h2 + (DEP ? c : 0u));
        uint32_t d = ask<16>(D, (size_t)((uint32_t)h3 & 0xFFFFFu) * 16);
        const uint64_t h4 = mix(h3 + (DEP ? d : 0u));
        uint32_t e = ask<4>(E, (size_t)((uint32_t)h4 & e_mask) * 4) + ask<4>(E, (size_t)((uint32_t)(h4 >> 32) & e_mask) * 4);
        acc += a + c + d + e;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
static uint8_t *g_buf;
static uint32_t *g_out;
template <typename F> static double time_ms(F launch, int rep)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < rep; r++) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / rep;
}
int main()
{
    const size_t bytes = (size_t)4 << 30;
    CK(hipMalloc((void **)&g_buf, bytes)); CK(hipMalloc((void **)&g_out, 64));
    CK(hipMemset(g_buf, 1, bytes));
    CK(hipDeviceSynchronize());
    const int wpcs[] = { 8, 16, 32 };
    const struct { const char *name; size_t foot; } feet[] = { { "128KB", (size_t)128 << 10 }, { "1MB", (size_t)1 << 20 }, { "4MB", (size_t)4 << 20 }, { "8MB", (size_t)8 << 20 },
                                                                  { "16MB", (size_t)16 << 20 }, { "32MB", (size_t)32 << 20 }, { "64MB", (size_t)64 << 20 }, { "128MB", (size_t)128 << 20 },
                                                                  { "4GB", (size_t)4 << 30 } };
    printf("{\n \"unit\": \"G lines/s (one aligned item = one line asked)\",\n \"rows\": [\n");
    bool first = true;
    for (int wpc : wpcs) {
        const dim3 grid(256 * wpc / 4), block(256);
        const uint64_t lanes = (uint64_t)grid.x * 256;
        for (auto &f : feet) {
            const uint32_t per_lane = 512;
            const double n = (double)lanes * per_lane;
#define ROW(KIND, WIDTH, MS) do { printf("%s  {\"kind\": \"%s\", \"bytes\": %d, \"footprint\": \"%s\", \"waves_per_cu\": %d, \"ms\": %.4f, \"g_lines_per_s\": %.2f}", first ? "" : ",\n", KIND, WIDTH, f.name, wpc, MS, n / (MS) * 1e-6); first = false; } while (0)
            double ms;
            ms = time_ms([&] { k_gather<4, 8><<<grid, block>>>(g_buf, (uint32_t)(f.foot / 4 - 1), per_lane, g_out); }, 3); ROW("gather", 4, ms);
            ms = time_ms([&] { k_gather<16, 8><<<grid, block>>>(g_buf, (uint32_t)(f.foot / 16 - 1), per_lane, g_out); }, 3); ROW("gather", 16, ms);
            ms = time_ms([&] { k_gather<32, 4><<<grid, block>>>(g_buf, (uint32_t)(f.foot / 32 - 1), per_lane, g_out); }, 3); ROW("gather", 32, ms);
            ms = time_ms([&] { k_gather<4, 1><<<grid, block>>>(g_buf, (uint32_t)(f.foot / 4 - 1), per_lane, g_out); }, 3); ROW("gather_1_in_flight", 4, ms);
            ms = time_ms([&] { k_chase<<<grid, block>>>(g_buf, (uint32_t)(f.foot / 4 - 1), per_lane, g_out); }, 3); ROW("chase", 4, ms);
#undef ROW
        }
        {
            const uint32_t per_lane = 64;
            const double n = (double)lanes * per_lane * 11;
            uint8_t *A = g_buf, *B = g_buf + ((size_t)1 << 20), *C = g_buf + ((size_t)16 << 20), *D = g_buf + ((size_t)32 << 20), *E = g_buf + ((size_t)64 << 20);
            double ms = time_ms([&] { k_mix<0><<<grid, block>>>(A, B, C, D, E, (1u << 24) - 1, per_lane, g_out); }, 3);       // E: 64 MB (a power of two below the 80)
            printf(",\n  {\"kind\": \"seed_mix_independent\", \"bytes\": 0, \"footprint\": \"128KB+1MB+16MB+16MB+64MB\", \"waves_per_cu\": %d, \"ms\": %.4f, \"g_lines_per_s\": %.2f}", wpc, ms, n / ms * 1e-6);
            ms = time_ms([&] { k_mix<1><<<grid, block>>>(A, B, C, D, E, (1u << 24) - 1, per_lane, g_out); }, 3);
            printf(",\n  {\"kind\": \"seed_mix_cascade\", \"bytes\": 0, \"footprint\": \"128KB+1MB+16MB+16MB+64MB\", \"waves_per_cu\": %d, \"ms\": %.4f, \"g_lines_per_s\": %.2f}", wpc, ms, n / ms * 1e-6);
        }
    }
    printf("\n ]\n}\n");
    return 0;
}
