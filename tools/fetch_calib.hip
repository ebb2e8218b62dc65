// tools/fetch_calib.hip - calibration of rocprofv3's FETCH_SIZE on gfx950 for the access widths this repo's kernels use
// (development aid, run on the GPU box by tools/fetch_calib.sh; not part of the product).
//   k_stream16   every lane reads 16 consecutive bytes, lanes consecutive (the guide's calibrated case: FETCH_SIZE = 1/2 of the bytes)
//   k_gather32   every lane reads one aligned 32-byte item at a random place (the seed kernel's wildcard-filter lines)
//   k_gather16   ... one aligned 16-byte item (pair-filter blocks)
//   k_gather4    ... one 4-byte word (bitmap / Bloom words)
// over a 4 GiB buffer (16 x the Infinity Cache), 64 Mi requests each: requested bytes are known exactly.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
__global__ void k_stream16(const uint4 *__restrict__ p, size_t n, uint32_t *out)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
template <int BYTES>
__global__ void k_gather(const uint8_t *__restrict__ p, size_t items, size_t nreq, uint32_t *out)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nreq; i += (size_t)gridDim.x * blockDim.x) {
        const size_t at = (size_t)(mix(i * 0x9E3779B97F4A7C15ull + 12345) % items) * BYTES;
        if (BYTES == 32) { const uint4 a = *(const uint4 *)(p + at), b = *(const uint4 *)(p + at + 16); acc += a.x ^ a.w ^ b.x ^ b.w; }
        else if (BYTES == 16) { const uint4 a = *(const uint4 *)(p + at); acc += a.x ^ a.w; }
        else acc += *(const uint32_t *)(p + at);
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main()
{
    const size_t bytes = (size_t)4 << 30, nreq = (size_t)64 << 20;
    uint8_t *buf; uint32_t *out;
    CK(hipMalloc((void **)&buf, bytes)); CK(hipMalloc((void **)&out, 64));
    CK(hipMemset(buf, 1, bytes));
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 2; rep++) {
        k_stream16<<<dim3(256 * 32), dim3(256)>>>((const uint4 *)buf, bytes / 16, out);
        k_gather<32><<<dim3(256 * 32), dim3(256)>>>(buf, bytes / 32, nreq, out);
        k_gather<16><<<dim3(256 * 32), dim3(256)>>>(buf, bytes / 16, nreq, out);
        k_gather<4><<<dim3(256 * 32), dim3(256)>>>(buf, bytes / 4, nreq, out);
    }
    CK(hipDeviceSynchronize());
    printf("stream16: %zu bytes; gathers: %zu requests of 32 / 16 / 4 bytes\n", bytes, nreq);
    return 0;
}
