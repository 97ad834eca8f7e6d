// Row-wise fully connected layers on the matrix cores (fp32 MFMA).
//
// Stands for the nn.Linear / 1x1 nn.Conv2d GEMMs that the reference's FCN and CRB2d blocks get
// from ATen (modules/layers/Blocks.py:9,14 and :35,39; used by modules/voxelnet/Pipe.py:9,
// VoxelNet.py:13 and modules/imhead/Pipe.py:88-92) together with their autograd:
//
//   forward : y[r][n] = [ReLU](sum_k x[r][k] * W[n][k] + b[n])          (+ BatchNorm statistics)
//   dgrad   : the same kernel with the weight read transposed (w_transposed = 1)
//   wgrad   : dW[n][k] = sum_r dz[r][n] * x[r][k]   (row strips -> slabs -> deterministic sum)
//
// Rows may carry weights (row_w): a compact row that stands for w identical rows of the
// reference's dense (V,35,C) tensor contributes w times to the BatchNorm sums.
// Matrices have explicit leading dimensions so layers can read/write column slices of the
// VFE concat buffers in place.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BM = 128, BN = 64, BK = 32, PITCH = BK + 4;

template <bool WT>
__global__ __launch_bounds__(256, 2) void linear_fwd(const float *__restrict__ x, int ldx,
                                                     const float *__restrict__ w, int ldw,
                                                     const float *__restrict__ bias, float *__restrict__ y,
                                                     int ldy, double *__restrict__ stats,
                                                     const float *__restrict__ row_w, long long R, int K, int N,
                                                     int relu) {
    __shared__ __attribute__((aligned(16))) float s_x[BM * PITCH];
    __shared__ __attribute__((aligned(16))) float s_w[BN * PITCH];
    __shared__ float s_red[4][2 * BN];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 31, lh = lane >> 5;
    const long long r0 = (long long)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;

    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const int a_base = (wv * 32 + li) * PITCH + 4 * lh;
    const int b_base0 = li * PITCH + 4 * lh, b_base1 = (32 + li) * PITCH + 4 * lh;

    for (int k0 = 0; k0 < K; k0 += BK) {
        __syncthreads();
#pragma unroll 4
        for (int u = 0; u < BM * BK / 256; ++u) {
            const int e = tid + 256 * u, r = e >> 5, k = e & 31;
            const long long gr = r0 + r;
            s_x[r * PITCH + k] = (gr < R && k0 + k < K) ? x[gr * ldx + k0 + k] : 0.f;
        }
#pragma unroll 4
        for (int u = 0; u < BN * BK / 256; ++u) {
            const int e = tid + 256 * u;
            int n, k;
            if (WT) { n = e & 63; k = e >> 6; } else { n = e >> 5; k = e & 31; }
            float v = 0.f;
            if (n0 + n < N && k0 + k < K)
                v = WT ? w[(long long)(k0 + k) * ldw + n0 + n] : w[(long long)(n0 + n) * ldw + k0 + k];
            s_w[n * PITCH + k] = v;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < BK / 8; ++q) {
            const float4 av = *(const float4 *)(s_x + a_base + 8 * q);
            const float4 b0 = *(const float4 *)(s_w + b_base0 + 8 * q);
            const float4 b1 = *(const float4 *)(s_w + b_base1 + 8 * q);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b0.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b1.x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b0.y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b1.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b0.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b1.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b0.w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b1.w, acc1, 0, 0, 0);
        }
    }

    const int c0 = n0 + li, c1 = c0 + 32;
    const float bias0 = (bias && c0 < N) ? bias[c0] : 0.f, bias1 = (bias && c1 < N) ? bias[c1] : 0.f;
    float s1a = 0.f, s2a = 0.f, s1b = 0.f, s2b = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const long long gr = r0 + wv * 32 + row;
        float v0 = acc0[r] + bias0, v1 = acc1[r] + bias1;
        if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
        if (gr < R) {
            const float rw = row_w ? row_w[gr] : 1.f;
            if (c0 < N) { y[gr * ldy + c0] = v0; s1a += rw * v0; s2a += rw * v0 * v0; }
            if (c1 < N) { y[gr * ldy + c1] = v1; s1b += rw * v1; s2b += rw * v1 * v1; }
        }
    }
    if (stats) {
        s1a += __shfl_xor(s1a, 32, 64); s2a += __shfl_xor(s2a, 32, 64);
        s1b += __shfl_xor(s1b, 32, 64); s2b += __shfl_xor(s2b, 32, 64);
        __syncthreads();
        if (lh == 0) {
            s_red[wv][li] = s1a; s_red[wv][32 + li] = s1b;
            s_red[wv][BN + li] = s2a; s_red[wv][BN + 32 + li] = s2b;
        }
        __syncthreads();
        if (tid < 2 * BN) {
            const int which = tid / BN, c = tid % BN;
            if (n0 + c < N) {
                const double t = (double)s_red[0][tid] + (double)s_red[1][tid] + (double)s_red[2][tid] +
                                 (double)s_red[3][tid];
                atomicAdd(stats + (size_t)which * N + n0 + c, t);
            }
        }
    }
}

// dW partial: slab[strip][n][k] over the rows of the strip.  Block = 64(n) x 64(k), wave (wn, wk).
constexpr int WR = 64;   // rows per LDS step
__global__ __launch_bounds__(256) void linear_wgrad(const float *__restrict__ x, int ldx,
                                                    const float *__restrict__ dz, int lddz,
                                                    float *__restrict__ slabs, long long R, int K, int N,
                                                    long long rows_per_strip) {
    __shared__ float s_z[WR * 64];
    __shared__ float s_x[WR * 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int wn = wv >> 1, wk = wv & 1;
    const int n0 = blockIdx.y * 64, k0 = blockIdx.z * 64;
    const long long rbeg = (long long)blockIdx.x * rows_per_strip;
    const long long rend = rbeg + rows_per_strip < R ? rbeg + rows_per_strip : R;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (long long rr = rbeg; rr < rend; rr += WR) {
        __syncthreads();
#pragma unroll 4
        for (int u = 0; u < WR * 64 / 256; ++u) {
            const int e = tid + 256 * u, r = e >> 6, c = e & 63;
            const long long gr = rr + r;
            const bool ok = gr < rend;
            s_z[e] = (ok && n0 + c < N) ? dz[gr * lddz + n0 + c] : 0.f;
            s_x[e] = (ok && k0 + c < K) ? x[gr * ldx + k0 + c] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int kk = 0; kk < WR / 2; ++kk) {
            const int row = 2 * kk + lh;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(s_z[row * 64 + wn * 32 + li], s_x[row * 64 + wk * 32 + li],
                                                      acc, 0, 0, 0);
        }
    }
    float *o = slabs + (size_t)blockIdx.x * N * K;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int k = k0 + wk * 32 + li;
        if (n < N && k < K) o[(size_t)n * K + k] = acc[r];
    }
}

__global__ void slab_reduce(const float *__restrict__ slabs, float *__restrict__ out, size_t per, int nslabs) {
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < per; e += (size_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < nslabs; ++k) s += slabs[(size_t)k * per + e];
        out[e] = s;
    }
}

inline long long strip_rows(long long R, int N, int K) {
    const long long blocks = (long long)mvx_cdiv(N, 64) * mvx_cdiv(K, 64);
    long long strips = 2048 / blocks;
    if (strips < 1) strips = 1;
    long long rows = (R + strips - 1) / strips;
    if (rows < 256) rows = 256;
    return ((rows + WR - 1) / WR) * WR;
}

}  // namespace

extern "C" int mvx_linear_forward(const float *x, int32_t ldx, const float *w, int32_t ldw, int32_t w_transposed,
                                  const float *bias, float *y, int32_t ldy, double *stats, const float *row_w,
                                  int64_t rows, int32_t k, int32_t n, int32_t relu, void *stream) {
    MVX_CHECK_ARG(x && w && y && rows >= 0 && k > 0 && n > 0 && ldx >= k && ldy >= n);
    MVX_CHECK_ARG(ldw >= (w_transposed ? n : k));
    hipStream_t st = (hipStream_t)stream;
    if (stats) {
        hipError_t e = hipMemsetAsync(stats, 0, sizeof(double) * 2 * n, st);
        if (e != hipSuccess) return (int)e;
    }
    if (rows == 0) return MVX_OK;
    const dim3 grid(mvx_cdiv(rows, BM), mvx_cdiv(n, BN));
    if (w_transposed)
        hipLaunchKernelGGL(linear_fwd<true>, grid, dim3(256), 0, st, x, ldx, w, ldw, bias, y, ldy, stats, row_w,
                           (long long)rows, k, n, relu);
    else
        hipLaunchKernelGGL(linear_fwd<false>, grid, dim3(256), 0, st, x, ldx, w, ldw, bias, y, ldy, stats, row_w,
                           (long long)rows, k, n, relu);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" size_t mvx_linear_wgrad_workspace_bytes(int64_t rows, int32_t k, int32_t n) {
    if (rows <= 0 || k <= 0 || n <= 0) return 256;
    const long long per = strip_rows(rows, n, k);
    const long long strips = (rows + per - 1) / per;
    return (size_t)strips * n * k * sizeof(float);
}

extern "C" int mvx_linear_wgrad(const float *x, int32_t ldx, const float *dz, int32_t lddz, float *dw,
                                int64_t rows, int32_t k, int32_t n, void *workspace, size_t workspace_bytes,
                                void *stream) {
    MVX_CHECK_ARG(x && dz && dw && workspace && rows >= 0 && k > 0 && n > 0 && ldx >= k && lddz >= n);
    hipStream_t st = (hipStream_t)stream;
    if (rows == 0) {
        hipError_t e = hipMemsetAsync(dw, 0, sizeof(float) * (size_t)n * k, st);
        return e == hipSuccess ? MVX_OK : (int)e;
    }
    const long long per = strip_rows(rows, n, k);
    const long long strips = (rows + per - 1) / per;
    MVX_CHECK_ARG(workspace_bytes >= (size_t)strips * n * k * sizeof(float));
    hipLaunchKernelGGL(linear_wgrad, dim3((unsigned)strips, mvx_cdiv(n, 64), mvx_cdiv(k, 64)), dim3(256), 0, st, x, ldx,
                       dz, lddz, (float *)workspace, (long long)rows, k, n, per);
    MVX_LAUNCH_CHECK();
    const size_t total = (size_t)n * k;
    hipLaunchKernelGGL(slab_reduce, dim3(mvx_cdiv(total, 256) > 1024 ? 1024 : mvx_cdiv(total, 256)), dim3(256), 0, st,
                       (const float *)workspace, dw, total, (int)strips);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}
