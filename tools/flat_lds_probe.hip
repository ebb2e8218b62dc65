// Development aid: does a generic (flat) pointer into dynamic LDS work through a noinline function on this GPU, and up to which offset?
// hipcc --offload-arch=gfx950 -O3 -o /tmp/flat_lds_probe tools/flat_lds_probe.hip && /tmp/flat_lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
struct Item { double k; unsigned i, pad; };
extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
__device__ __attribute__((noinline)) void touch(Item *p, int n, double seed)
{
    for (int i = 0; i < n; i++) { Item t; t.k = seed + i; t.i = (unsigned)i; t.pad = 0; p[i] = t; }
    for (int i = 1; i < n; i++) { Item v = p[i]; int j = i; while (j > 0 && p[j - 1].k < v.k) { p[j] = p[j - 1]; j--; } p[j] = v; }
}
__global__ void k(double *out, int n, int stride, int use_lds, Item *g)
{
    Item *p = use_lds ? (Item *)(smem + (size_t)threadIdx.x * stride) : g + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * n;
    touch(p, n, (double)threadIdx.x);
    out[blockIdx.x * blockDim.x + threadIdx.x] = p[0].k;
}
int main()
{
    double *out; Item *g;
    hipMalloc(&out, 8 * 1024); hipMalloc(&g, sizeof(Item) * 1024 * 96);
    for (int use = 0; use < 2; use++)
        for (int tpb : {32, 64, 128}) {
            const int n = 16, stride = n * 16 + 16;
            const size_t lds = (size_t)tpb * stride;
            hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            k<<<dim3(4), dim3(tpb), lds>>>(out, n, stride, use, g);
            hipError_t e = hipDeviceSynchronize();
            double h[4]; hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
            printf("use_lds %d tpb %d lds %zu: %s first %.1f %.1f\n", use, tpb, lds, hipGetErrorString(e), h[0], h[1]);
        }
    return 0;
}
