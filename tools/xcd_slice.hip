// tools/xcd_slice.hip - does a table that is asked at scattered places run at the L2's rate when every XCD is only ever asked for ITS
// eighth of it?  (Development aid for the XCD-sliced filter cascade of the seed kernel, DESIGN 5.6 / VERDICT r05 item 2; run on the GPU
// box by tools/xcd_slice.sh; not part of the product.)
//   k_xcc             which XCD a workgroup runs on (s_getreg_b32 HW_REG_XCC_ID), by blockIdx: the dispatcher's round robin
//   k_gather<B, 0>    every lane asks aligned items of B bytes at hashed places of the WHOLE table (tools/gather_ceiling.hip)
//   k_gather<B, 1>    ... of the slice that belongs to the XCD its workgroup runs on: table / 8, slice number = XCC_ID
//   k_list<B>         the cascade's second pass as it would be built: items (12 bytes: place + payload) are read from the XCD's own list,
//                     coalesced, and each one asks the table at its place - the list is a stream, the table a gather
// One JSON document on stdout.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
__device__ __forceinline__ uint32_t xcc_id() { uint32_t x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); return x & 15u; }
template <int BYTES> __device__ __forceinline__ uint32_t ask(const uint8_t *__restrict__ p, size_t at)
{
    if (BYTES == 32) { const uint4 a = *(const uint4 *)(p + at), b = *(const uint4 *)(p + at + 16); return a.x ^ a.w ^ b.x ^ b.w; }
    if (BYTES == 16) { const uint4 a = *(const uint4 *)(p + at); return a.x ^ a.w; }
    return *(const uint32_t *)(p + at);
}
__global__ void k_xcc(uint32_t *out) { if (threadIdx.x == 0) out[blockIdx.x] = xcc_id(); }
template <int BYTES, int SLICED, int UNROLL>
__global__ void k_gather(const uint8_t *__restrict__ p, uint32_t item_mask, uint32_t per_lane, uint32_t *out)
{
    uint32_t acc = 0;
    const uint64_t lane = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t base = SLICED ? (size_t)xcc_id() * ((size_t)item_mask + 1) * BYTES : 0;   // (sliced: item_mask spans ONE slice)
    for (uint32_t i = 0; i < per_lane; i += UNROLL) {
        uint32_t v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = ask<BYTES>(p, base + (size_t)((uint32_t)mix((lane << 20) + i + u) & item_mask) * BYTES);
#pragma unroll
        for (int u = 0; u < UNROLL; u++) acc += v[u];
    }
    if (acc == 0x12345678u) out[0] = acc;
}
// lists: list x holds n items of 12 bytes (3 words: place inside the slice, two words of payload); the workgroups of XCD x take blocks of
// 256 items from a counter of their list
template <int BYTES>
__global__ void k_list(const uint8_t *__restrict__ p, const uint32_t *__restrict__ lists, uint32_t per_list, uint32_t slice_items, uint32_t *counters, uint32_t *out)
{
    const uint32_t x = xcc_id() & 7u;
    const uint32_t *L = lists + (size_t)x * per_list * 3;
    const size_t base = (size_t)x * slice_items * BYTES;
    __shared__ uint32_t blk;
    uint32_t acc = 0;
    for (;;) {
        if (threadIdx.x == 0) blk = atomicAdd(&counters[x * 32], 1u);
        __syncthreads();
        const uint32_t b = blk;
        __syncthreads();
        if ((uint64_t)b * 1024 >= per_list) break;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t i = b * 1024 + u * 256 + threadIdx.x;
            if (i < per_list) {
                const uint32_t at = L[(size_t)i * 3], w1 = L[(size_t)i * 3 + 1], w2 = L[(size_t)i * 3 + 2];
                acc += ask<BYTES>(p, base + (size_t)(at & (slice_items - 1)) * BYTES) + w1 + w2;
            }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}
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
    uint8_t *buf; uint32_t *out, *lists, *counters;
    const size_t bytes = (size_t)256 << 20;
    CK(hipMalloc((void **)&buf, bytes)); CK(hipMalloc((void **)&out, 1 << 20)); CK(hipMalloc((void **)&counters, 4096));
    CK(hipMemset(buf, 1, bytes));
    printf("{\n");
    {   // the dispatcher's round robin
        const int nb = 4096;
        k_xcc<<<nb, 64>>>(out); CK(hipDeviceSynchronize());
        std::vector<uint32_t> h(nb);
        CK(hipMemcpy(h.data(), out, nb * 4, hipMemcpyDeviceToHost));
        int same = 0, hist[16] = {0};
        for (int i = 0; i < nb; i++) { same += (h[i] == (uint32_t)(i % 8)); hist[h[i] & 15]++; }
        printf(" \"xcc_of_block\": {\"blocks\": %d, \"xcc_equals_block_mod_8\": %d, \"first_16\": [", nb, same);
        for (int i = 0; i < 16; i++) printf("%s%u", i ? ", " : "", h[i]);
        printf("], \"blocks_per_xcc\": [");
        for (int i = 0; i < 8; i++) printf("%s%d", i ? ", " : "", hist[i]);
        printf("]},\n");
    }
    printf(" \"unit\": \"G lines/s (one aligned item = one line asked)\",\n \"rows\": [\n");
    bool first = true;
    const int wpc = 16;
    const dim3 grid(256 * wpc / 4), block(256);
    const uint64_t lanes = (uint64_t)grid.x * 256;
    const uint32_t per_lane = 512;
    const double n = (double)lanes * per_lane;
    const struct { const char *name; size_t foot; } feet[] = { { "2MB", (size_t)2 << 20 }, { "4MB", (size_t)4 << 20 }, { "16MB", (size_t)16 << 20 }, { "32MB", (size_t)32 << 20 }, { "64MB", (size_t)64 << 20 } };
#define ROW(KIND, WIDTH, FOOT, MS, N) do { printf("%s  {\"kind\": \"%s\", \"bytes\": %d, \"footprint\": \"%s\", \"waves_per_cu\": %d, \"ms\": %.4f, \"g_lines_per_s\": %.2f}", first ? "" : ",\n", KIND, WIDTH, FOOT, wpc, MS, (N) / (MS) * 1e-6); first = false; } while (0)
    for (auto &f : feet) {
        double ms;
        ms = time_ms([&] { k_gather<4, 0, 8><<<grid, block>>>(buf, (uint32_t)(f.foot / 4 - 1), per_lane, out); }, 3); ROW("whole_table", 4, f.name, ms, n);
        ms = time_ms([&] { k_gather<16, 0, 8><<<grid, block>>>(buf, (uint32_t)(f.foot / 16 - 1), per_lane, out); }, 3); ROW("whole_table", 16, f.name, ms, n);
        ms = time_ms([&] { k_gather<32, 0, 4><<<grid, block>>>(buf, (uint32_t)(f.foot / 32 - 1), per_lane, out); }, 3); ROW("whole_table", 32, f.name, ms, n);
        if (f.foot >= ((size_t)16 << 20)) {
            ms = time_ms([&] { k_gather<4, 1, 8><<<grid, block>>>(buf, (uint32_t)(f.foot / 8 / 4 - 1), per_lane, out); }, 3); ROW("slice_per_xcd", 4, f.name, ms, n);
            ms = time_ms([&] { k_gather<16, 1, 8><<<grid, block>>>(buf, (uint32_t)(f.foot / 8 / 16 - 1), per_lane, out); }, 3); ROW("slice_per_xcd", 16, f.name, ms, n);
            ms = time_ms([&] { k_gather<32, 1, 4><<<grid, block>>>(buf, (uint32_t)(f.foot / 8 / 32 - 1), per_lane, out); }, 3); ROW("slice_per_xcd", 32, f.name, ms, n);
        }
    }
    {   // the second pass of the cascade: 150 M items (1 M reads x 150 wildcard asks) in eight lists, 16 MB table of 32-byte lines; and 77 M items on 16-byte blocks
        const uint32_t per_list = 150000000u / 8;
        CK(hipMalloc((void **)&lists, (size_t)per_list * 8 * 12));
        std::vector<uint32_t> h((size_t)1 << 22);
        for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(((i * 0x9E3779B97F4A7C15ull) >> 20) & 0xFFFFFFFFu);
        for (size_t o = 0; o < (size_t)per_list * 8 * 3; o += h.size()) CK(hipMemcpy(lists + o, h.data(), std::min(h.size(), (size_t)per_list * 8 * 3 - o) * 4, hipMemcpyHostToDevice));
        double ms;
        ms = time_ms([&] { CK(hipMemsetAsync(counters, 0, 4096)); k_list<32><<<grid, block>>>(buf, lists, per_list, (uint32_t)(((size_t)16 << 20) / 8 / 32), counters, out); }, 3);
        ROW("lists_of_12_byte_items_then_slice_per_xcd", 32, "16MB", ms, (double)per_list * 8);
        const uint32_t per_list2 = 77000000u / 8;
        ms = time_ms([&] { CK(hipMemsetAsync(counters, 0, 4096)); k_list<16><<<grid, block>>>(buf, lists, per_list2, (uint32_t)(((size_t)16 << 20) / 8 / 16), counters, out); }, 3);
        ROW("lists_of_12_byte_items_then_slice_per_xcd", 16, "16MB", ms, (double)per_list2 * 8);
    }
    printf("\n ]\n}\n");
    return 0;
}
