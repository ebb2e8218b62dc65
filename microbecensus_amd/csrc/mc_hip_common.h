// mc_hip_common.h - what every kernel file of libmcensus_hip.so shares: includes, the error string, the device counters, and the
// wave-level helpers (lane id, DPP prefix sum, wave-scope barrier, slot allocation with one global atomic per wave / workgroup).
// One translation unit: mc_hip.hip includes the kernel files in pipeline order (see its header).
#pragma once
#include <cstddef>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mcensus.h"
#include "mc_finish.h"
#include "mc_index.h"

static_assert(sizeof(McRow) % 8 == 0 && sizeof(McRow) == sizeof(mc_row) && offsetof(McRow, ident) == offsetof(mc_row, ident) && offsetof(McRow, loge) == offsetof(mc_row, loge) &&
                  offsetof(McRow, score) == offsetof(mc_row, score) && offsetof(McRow, frame) == offsetof(mc_row, nmatch),
              "the device row is handed out as the ABI row");

static thread_local std::string g_err;
// MC_OPEN_TIMING in the environment: where the time of opening an engine and of its first run goes (stderr; development aid)
static double mc_now() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
static bool mc_open_timing() { static const bool on = getenv("MC_OPEN_TIMING") != nullptr; return on; }
#define MC_OT(label, t0) do { if (mc_open_timing()) { const double now_ = mc_now(); fprintf(stderr, "open-timing %-28s %8.1f ms\n", label, (now_ - (t0)) * 1e3); (t0) = now_; } } while (0)
extern "C" const char *mc_last_error(void) { return g_err.c_str(); }

#define HIPCK(call)                                                                                         \
    do {                                                                                                    \
        hipError_t e_ = (call);                                                                             \
        if (e_ != hipSuccess) {                                                                             \
            g_err = std::string(#call) + ": " + hipGetErrorString(e_);                                       \
            return -1;                                                                                      \
        }                                                                                                   \
    } while (0)

enum { C_TASKS = 0, C_GAPS, C_HSPS, C_HEADS, C_ROWS, C_OVERFLOW, C_SEGS, C_BEST, C_RETRY, C_HEAVY, C_HEAVY2, C_ITEMS, C_RETRY2, C_HEAVY3, C_LIGHT0, C_LIGHT1, C_LIGHT2, C_LIGHT3, C_HSPS2, C_HPAD, C_GPAD, C_ORDER, C_ORDER2, C_ORDER3, C_OTAKE, C_OTAKE2, C_OTAKE3, C_HEAVY1, C_ENCHUNK, C_EVCHUNK, C_GTAKE, C_GTAKE2, C_N = 32 };
enum { S_LOOKUPS = 0, S_KEYPROBES, S_TASKS, S_EXACT = 16, S_WILD, S_PAIRS, S_PROBES, S_N = 20 };   // 64-bit algorithmic-traffic counters of k_enumerate; slots 4..: cycle counters of the MC_EXP_TIMING build

extern __shared__ __attribute__((aligned(16))) uint8_t mc_smem[];   // dynamic LDS of the kernels that use it

__device__ __forceinline__ int mc_lane() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
// orders the wave's own LDS traffic for the compiler; the hardware executes one wave's LDS instructions in order
// inclusive prefix sum over the 64 lanes in six DPP additions: shifts inside the rows of 16, then the row totals carried across
__device__ __forceinline__ uint32_t mc_wave_scan_add(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);    // row_shr:1 (lanes shifted in from outside a row read 0)
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);    // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);    // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);    // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);   // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);   // row_bcast:31 into rows 2 and 3
    return v;
}
// Streaming records (read once, or written once for a later kernel) go past the caches' keep lists - "nontemporal" loads and stores -,
// so that the lines a gather kernel comes back to (the subjects' residues in k_eval_seeds: 3.7 MB, an L2's worth) stay in the L2.
// Round 5, per 1 M reads of 150 bp: k_eval_seeds 2.50 -> 2.45 ms with the loads of the seed hits alone, 2.45 with the stores of HSPs
// and gap tasks alone, 2.40 with both; the seed kernel's own stores of the hits: 6.12 -> 6.14 (left as plain stores - its L2 misses
// are index lines by the hundred per read, the hits are 3 % of its fills).  T: a record of 4-byte words.
typedef uint32_t mc_u32x4 __attribute__((ext_vector_type(4)));
template <typename T> __device__ __forceinline__ void mc_store_stream(T *dst, const T &v)
{
    static_assert(sizeof(T) % 4 == 0, "records of 4-byte words");
    uint32_t w[sizeof(T) / 4];
    __builtin_memcpy(w, &v, sizeof(T));
    uint32_t *d = (uint32_t *)dst;
    if (sizeof(T) % 16 == 0) {
#pragma unroll
        for (unsigned i = 0; i < sizeof(T) / 16; i++) { mc_u32x4 x = { w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3] }; __builtin_nontemporal_store(x, (mc_u32x4 *)(d + 4 * i)); }
    } else {
#pragma unroll
        for (unsigned i = 0; i < sizeof(T) / 4; i++) __builtin_nontemporal_store(w[i], d + i);
    }
}
template <typename T> __device__ __forceinline__ T mc_load_stream(const T *src)
{
    static_assert(sizeof(T) % 16 == 0, "records of 16-byte words");
    T v;
    uint32_t w[sizeof(T) / 4];
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 16; i++) { const mc_u32x4 x = __builtin_nontemporal_load((const mc_u32x4 *)src + i); w[4 * i] = x.x; w[4 * i + 1] = x.y; w[4 * i + 2] = x.z; w[4 * i + 3] = x.w; }
    __builtin_memcpy(&v, w, sizeof(T));
    return v;
}
__device__ __forceinline__ void mc_wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

// one atomic per wave: the lanes with want == true receive consecutive slots of a global counter
__device__ __forceinline__ uint32_t mc_wave_alloc(uint32_t *counter, bool want)
{
    const unsigned long long m = __ballot(want);
    if (m == 0) return 0;
    const int lane = mc_lane(), leader = __builtin_ctzll(m);
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
    return base + (uint32_t)__popcll(m & ((1ull << lane) - 1));
}

// one atomic per 256-thread workgroup (every thread of the block must call it): a device-scope atomic on ONE counter runs at
// the memory side at ~125 M/s, so even one per wave is too many for kernels of millions of threads
__device__ __forceinline__ uint32_t mc_block_alloc(uint32_t *counter, bool want)
{
    __shared__ uint32_t wcnt[4], wbase[4];
    const unsigned long long m = __ballot(want);
    const int lane = mc_lane(), wv = threadIdx.x >> 6;
    if (lane == 0) wcnt[wv] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t c0 = wcnt[0], c1 = wcnt[1], c2 = wcnt[2], c3 = wcnt[3], tot = c0 + c1 + c2 + c3;
        const uint32_t b = tot ? atomicAdd(counter, tot) : 0u;
        wbase[0] = b; wbase[1] = b + c0; wbase[2] = b + c0 + c1; wbase[3] = b + c0 + c1 + c2;
    }
    __syncthreads();
    const uint32_t r = wbase[wv] + (uint32_t)__popcll(m & ((1ull << lane) - 1));
    __syncthreads();                                             // (the arrays are reused by the next call)
    return r;
}

// ... and slots of up to 8 counters at once (every thread of the 256-thread block calls it; bit k of `wants`: the thread takes a slot of
// counters[idx[k]], returned in off[k]): the N atomics of a workgroup travel together - called one after the other every one of
// them is a trip to the memory side that the whole workgroup waits for at a barrier (k_heavy_lists made eight of them).
template <int N>
__device__ __forceinline__ void mc_block_alloc_multi(uint32_t *counters, const int *idx, uint32_t wants, uint32_t (&off)[N])
{
    static_assert(N <= 8, "eight counters at most");
    __shared__ uint32_t wcnt[4][8], wbase[4][8];
    const int lane = mc_lane(), wv = threadIdx.x >> 6;
    const unsigned long long lt = (1ull << lane) - 1;
    uint32_t rank[N];
#pragma unroll
    for (int k = 0; k < N; k++) {
        const unsigned long long m = __ballot((wants >> k) & 1u);
        rank[k] = (uint32_t)__popcll(m & lt);
        if (lane == 0) wcnt[wv][k] = (uint32_t)__popcll(m);
    }
    __syncthreads();
    if (threadIdx.x < N) {
        const int k = threadIdx.x;
        const uint32_t c0 = wcnt[0][k], c1 = wcnt[1][k], c2 = wcnt[2][k], c3 = wcnt[3][k], tot = c0 + c1 + c2 + c3;
        const uint32_t b = tot ? atomicAdd(&counters[idx[k]], tot) : 0u;
        wbase[0][k] = b; wbase[1][k] = b + c0; wbase[2][k] = b + c0 + c1; wbase[3][k] = b + c0 + c1 + c2;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; k++) off[k] = wbase[wv][k] + rank[k];
    __syncthreads();                                             // (the arrays are reused by the next call)
}

// copies the hot members of the tables into LDS (block-wide; callers __syncthreads() afterwards)
__device__ __forceinline__ void mc_load_hot(McHot *H, const McTables *T)
{
    for (int i = threadIdx.x; i < 32 * 32 / 4; i += blockDim.x) ((uint32_t *)H->sub)[i] = ((const uint32_t *)T->sub)[i];
    if (threadIdx.x < 32) H->grp[threadIdx.x] = T->grp[threadIdx.x];
    if (threadIdx.x == 0) { H->xdrop_ungapped = T->xdrop_ungapped; H->xdrop_gapped = T->xdrop_gapped; H->gap_trigger = T->gap_trigger; }
}
