// Snapshot of the ROUND-1 linear_fwd kernel text with knock-out knobs (times only); generated then by a script that tracked the kernel source.
//   knob 1: barriers only in the first chunk   2: LDS tile stores only in the first chunk   4: no global prefetch loads
//        8: no output stores / statistics
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define MVX_REP 32
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int BM = 128, BK = 32;
constexpr int PITCH = BK + 4;
namespace {
template <bool WT, int NT, bool VEC, int KNOB>
__global__ __launch_bounds__(256) void linear_fwd(const float *__restrict__ x, int ldx, const float *__restrict__ w,
                                                  int ldw, const float *__restrict__ bias, float *__restrict__ y,
                                                  int ldy, double *__restrict__ stats, const float *__restrict__ row_w,
                                                  long long R, int K, int N, int relu, int k_per_split,
                                                  unsigned *__restrict__ done_counter, double fin_count, double fin_eps,
                                                  float *__restrict__ fin_mean_inv) {
    constexpr int BNL = 32 * NT;
    constexpr int XV = BM * BK / 4 / 256;          // float4 per thread for the x tile (4)
    constexpr int WV = BNL * BK / 4 / 256;         // float4 per thread for the w tile (NT)
    __shared__ __attribute__((aligned(16))) float s_x[BM * PITCH];
    __shared__ __attribute__((aligned(16))) float s_w[BNL * PITCH];
    __shared__ float s_red[4][2 * BNL];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 31, lh = lane >> 5;
    // column block fastest: the workgroups that share an x tile run together and read it from HBM once
    const long long r0 = (long long)blockIdx.y * BM;
    const int n0 = blockIdx.x * BNL;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int a_base = (wv * 32 + li) * PITCH + 4 * lh;
    const int b_base = li * PITCH + 4 * lh;

    // Prefetch registers.  Loads are UNCONDITIONAL from a clamped (always valid) address: rows >= R and columns
    // >= N only feed outputs that the epilogue drops, so whatever is loaded there is harmless; only the K tail
    // (k >= K inside the last chunk) must be zero, and that is done when the LAST chunk is written to LDS.
    // (A conditional load merged with a zero made the compiler wait for the load right where it was issued --
    // s_waitcnt vmcnt(0) inside the prefetch -- exposing the global latency once per chunk.)
    f32x4 xr[XV], wr[WV];
    auto load_tiles = [&](int k0) __attribute__((always_inline)) {
        if (VEC) {
#pragma unroll
            for (int u = 0; u < XV; ++u) {
                const int c = tid + 256 * u, r = c >> 3, part = c & 7;
                const long long gr = r0 + r;
                const bool ok = gr < R && k0 + part * 4 < K;
                xr[u] = *(const f32x4 *)(ok ? x + gr * ldx + k0 + part * 4 : x);
            }
#pragma unroll
            for (int u = 0; u < WV; ++u) {
                const int c = tid + 256 * u;
                if (!WT) {
                    const int n = c >> 3, part = c & 7;
                    const bool ok = n0 + n < N && k0 + part * 4 < K;
                    wr[u] = *(const f32x4 *)(ok ? w + (long long)(n0 + n) * ldw + k0 + part * 4 : w);
                } else {
                    // transposed weight: lanes run along n (coalesced 4-byte loads), a thread collects 4 consecutive
                    // k of its column so that the LDS image [n][k] is written with one 16-byte store (a float4 load
                    // along n would have to be scattered into 4 rows: 16-way bank conflicts)
                    const int n = c % BNL, kg = c / BNL;
                    f32x4 kk;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bool ok = k0 + kg * 4 + j < K && n0 + n < N;
                        kk[j] = *(ok ? w + (long long)(k0 + kg * 4 + j) * ldw + n0 + n : w);
                    }
                    wr[u] = kk;
                }
            }
        }
    };
    auto store_tiles = [&](int k0) __attribute__((always_inline)) {
        if (VEC) {
            const bool tail = k0 + BK > K;            // block-uniform: only the last chunk can hold k >= K
#pragma unroll
            for (int u = 0; u < XV; ++u) {
                const int c = tid + 256 * u;
                f32x4 v = xr[u];
                if (tail && k0 + (c & 7) * 4 >= K) v = f32x4{0.f, 0.f, 0.f, 0.f};
                *(f32x4 *)(s_x + (c >> 3) * PITCH + (c & 7) * 4) = v;
            }
#pragma unroll
            for (int u = 0; u < WV; ++u) {
                const int c = tid + 256 * u;
                f32x4 v = wr[u];
                if (!WT) {
                    if (tail && k0 + (c & 7) * 4 >= K) v = f32x4{0.f, 0.f, 0.f, 0.f};
                    *(f32x4 *)(s_w + (c >> 3) * PITCH + (c & 7) * 4) = v;
                } else {
                    const int n = c % BNL, kg = c / BNL;
                    if (tail) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (k0 + kg * 4 + j >= K) v[j] = 0.f;
                    }
                    *(f32x4 *)(s_w + n * PITCH + kg * 4) = v;
                }
            }
        } else {
            // scalar path (leading dimension not a multiple of 4): no prefetch
            for (int e = tid; e < BM * BK; e += 256) {
                const int r = e >> 5, k = e & 31;
                const long long gr = r0 + r;
                s_x[r * PITCH + k] = (gr < R && k0 + k < K) ? x[gr * ldx + k0 + k] : 0.f;
            }
            for (int e = tid; e < BNL * BK; e += 256) {
                int n, k;
                if (WT) { n = e % BNL; k = e / BNL; } else { n = e >> 5; k = e & 31; }
                float v = 0.f;
                if (n0 + n < N && k0 + k < K)
                    v = WT ? w[(long long)(k0 + k) * ldw + n0 + n] : w[(long long)(n0 + n) * ldw + k0 + k];
                s_w[n * PITCH + k] = v;
            }
        }
    };

    // split-K: blockIdx.z owns k in [kbeg, kend) and writes its partial product to slab z of y
    const int kbeg = blockIdx.z * k_per_split, kend = min(K, kbeg + k_per_split);
    y += (size_t)blockIdx.z * (size_t)R * ldy;
    load_tiles(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        if (!(KNOB & 1) || k0 == kbeg) __syncthreads();
        if (!(KNOB & 2) || k0 == kbeg) store_tiles(k0);
        if (!(KNOB & 1) || k0 == kbeg) __syncthreads();
        if (k0 + BK < kend) { if (!(KNOB & 4)) load_tiles(k0 + BK); }
#pragma unroll
        for (int q = 0; q < BK / 8; ++q) {
            const float4 av = *(const float4 *)(s_x + a_base + 8 * q);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float4 bv = *(const float4 *)(s_w + b_base + t * 32 * PITCH + 8 * q);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[t], 0, 0, 0);
            }
        }
    }

    float s1[NT], s2[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int c = n0 + t * 32 + li;
        const float bs = (bias && c < N) ? bias[c] : 0.f;
        s1[t] = 0.f; s2[t] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const long long gr = r0 + wv * 32 + row;
            float v = acc[t][r] + bs;
            if (relu) v = fmaxf(v, 0.f);
            if (gr < R && c < N) {
                if (!(KNOB & 8) || v == 12345.678f) y[gr * ldy + c] = v;
                const float rw = row_w ? row_w[gr] : 1.f;
                s1[t] += rw * v;
                s2[t] += rw * v * v;
            }
        }
    }
    if (stats && (!(KNOB & 8) || s1[0] == 12345.678f)) {
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float a = s1[t] + __shfl_xor(s1[t], 32, 64), b = s2[t] + __shfl_xor(s2[t], 32, 64);
            if (lh == 0) { s_red[wv][t * 32 + li] = a; s_red[wv][BNL + t * 32 + li] = b; }
        }
        __syncthreads();
        for (int e = tid; e < 2 * BNL; e += 256) {
            const int which = e / BNL, c = e % BNL;
            if (n0 + c < N) {
                const double t = (double)s_red[0][e] + (double)s_red[1][e] + (double)s_red[2][e] + (double)s_red[3][e];
                atomicAdd(stats + ((size_t)(blockIdx.y % MVX_REP) * 2 + which) * N + n0 + c, t);
            }
        }
    }
}

}

template <int KNOB> void run(const char *what, float *x, float *w, float *b, float *y, double *stats, int R, int K, int N) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid((N + 127) / 128, (R + BM - 1) / BM, 1);
    hipLaunchKernelGGL((linear_fwd<false, 4, true, KNOB>), grid, dim3(256), 0, 0, x, K, w, K, b, y, N, stats, (const float *)nullptr, (long long)R, K, N, 1, K, (unsigned *)nullptr, 0.0, 0.0, (float *)nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((linear_fwd<false, 4, true, KNOB>), grid, dim3(256), 0, 0, x, K, w, K, b, y, N, stats, (const float *)nullptr, (long long)R, K, N, 1, K, (unsigned *)nullptr, 0.0, 0.0, (float *)nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("knob %2d  %-44s %.3f ms  %.1f TFLOP/s-equivalent\n", KNOB, what, ms, 2.0 * R * K * N / ms / 1e9);
}
__global__ void fill_random(float *p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (float)(h >> 8) * (1.0f / 16777216.0f) - 0.5f;
    }
}
int main() {
    const int R = 20000, K = 768, N = 768;
    float *x, *w, *b, *y; double *stats;
    hipMalloc(&x, (size_t)R * K * 4); hipMalloc(&w, (size_t)N * K * 4); hipMalloc(&b, N * 4); hipMalloc(&y, (size_t)R * N * 4);
    hipMalloc(&stats, 8 * 32 * 2 * N);
    fill_random<<<4096, 256>>>(x, (size_t)R * K, 1u); fill_random<<<256, 256>>>(w, (size_t)N * K, 2u); hipMemset(b, 0, N * 4);
    hipMemset(stats, 0, 8 * 32 * 2 * N);
    run<0>("production", x, w, b, y, stats, R, K, N);
    run<1>("barriers only in chunk 0", x, w, b, y, stats, R, K, N);
    run<2>("LDS stores only in chunk 0", x, w, b, y, stats, R, K, N);
    run<4>("no global prefetch", x, w, b, y, stats, R, K, N);
    run<6>("no prefetch, no LDS stores", x, w, b, y, stats, R, K, N);
    run<7>("MFMA + LDS reads only", x, w, b, y, stats, R, K, N);
    run<8>("production without the epilogue stores", x, w, b, y, stats, R, K, N);
    run<15>("MFMA + LDS reads, no epilogue", x, w, b, y, stats, R, K, N);
    return 0;
}
