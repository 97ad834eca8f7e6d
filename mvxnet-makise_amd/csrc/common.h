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

// ---- internal helpers shared between translation units (NOT part of the C ABI) -----------------------
// conv3d.hip: compacted (plane, tile) step lists of the background-aware weight gradient and its closed-form term
int mvxi_wgrad_step_list(const int32_t *in_halo_flags, int din, int dout, int ntiles, int stride_d, int pad_d, int *list,
                         int *count, hipStream_t st);
int mvxi_wgrad_rank1(const float *tap_sums, const float *c_in, float *dw, int din, int dout, int cin, int cout, int stride_d,
                     int pad_d, hipStream_t st);

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

// ---- BatchNorm finalisation by the last workgroup of the producing kernel ----------------------------
// Every workgroup calls this after its statistics atomics (all threads, block-uniform arguments).  The workgroup that
// arrives last turns the replicated sums into mean and 1/sqrt(var + eps), which saves the separate mvx_bn_finalize
// launch.  Only device-scope ATOMICS carry the protocol (sums, counter, and the final reads), all served by the same
// coherence point, so no agent-scope fence is needed: such a fence writes the XCD's L2 back (the output tile each
// workgroup has just stored) and cost 13 % of the step when it was tried.  What the protocol does need is that every
// thread's statistics atomics have been PERFORMED at L2 before thread 0 bumps the counter: a workgroup-scope release
// emits no wait on gfx950 (the round-1 binary had the non-returning global_atomic_add_f64 followed by s_barrier and the
// counter atomic with nothing in between), so each thread drains its own vector-memory counter explicitly
// (mvx_drain_vmem: `s_waitcnt vmcnt(0)`; non-returning atomics are vmcnt-tracked on gfx9 and acknowledged by L2 once
// performed) before the barrier.  No cache write-back is involved.  `s_flag` is one int of LDS.
__device__ __forceinline__ void mvx_drain_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ void bn_finalize_by_last_block(unsigned *done_counter, unsigned total_blocks,
                                                          double *stats, int C, double count, double eps,
                                                          float *mean_inv, int *s_flag) {
    mvx_drain_vmem();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned prev = atomicAdd(done_counter, 1u);
        *s_flag = (prev == total_blocks - 1u);
    }
    __syncthreads();
    if (!*s_flag) return;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        double v1[MVX_REP], v2[MVX_REP];
#pragma unroll
        for (int rp = 0; rp < MVX_REP; ++rp) {            // 64 independent device-scope reads in flight
            v1[rp] = __hip_atomic_load(stats + ((size_t)rp * 2) * C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v2[rp] = __hip_atomic_load(stats + ((size_t)rp * 2 + 1) * C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int rp = 0; rp < MVX_REP; ++rp) { s1 += v1[rp]; s2 += v2[rp]; }
        const double mean = s1 / count;
        double var = s2 / count - mean * mean;
        if (var < 0.0) var = 0.0;
        mean_inv[c] = (float)mean;
        mean_inv[C + c] = (float)(1.0 / sqrt(var + eps));
    }
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
