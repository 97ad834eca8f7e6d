// Shared device/host helpers for libmvx_hip (gfx950 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mvx_hip.h"

#define MVX_WAVE 64
#define MVX_REP MVX_STATS_REPLICAS

#define MVX_CHECK_ARG(cond)            \
    do {                               \
        if (!(cond)) return MVX_EINVAL; \
    } while (0)

#define MVX_LAUNCH_CHECK()                              \
    do {                                                \
        hipError_t e__ = hipGetLastError();             \
        if (e__ != hipSuccess) return (int)e__;         \
    } while (0)

static inline unsigned mvx_cdiv(long long a, long long b) { return (unsigned)((a + b - 1) / b); }

// ---- wave / block reductions and scans -------------------------------------------------
__device__ __forceinline__ int wave_incl_scan_i32(int v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// Exclusive scan across a 1024-thread block (16 waves).  `smem` needs 17 ints.  Returns the
// exclusive prefix of this thread's value; *total receives the block sum.
__device__ __forceinline__ int block_excl_scan_i32(int v, int *smem, int *total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int inc = wave_incl_scan_i32(v);
    if (lane == 63) smem[wid] = inc;
    __syncthreads();
    if (wid == 0) {
        int w = lane < nw ? smem[lane] : 0;
        int wi = wave_incl_scan_i32(w);
        if (lane < nw) smem[lane] = wi - w;
        if (lane == nw - 1) smem[16] = wi;
    }
    __syncthreads();
    int res = inc - v + smem[wid];
    *total = smem[16];
    __syncthreads();
    return res;
}

__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
    return v;
}
