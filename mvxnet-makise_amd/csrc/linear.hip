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
//
// Forward tile: 128 rows x (32*NT) columns per workgroup (NT = 2 or 4), wave w owns 32 rows x all
// columns; K in chunks of 32 staged through LDS (rows padded to 36 floats, one ds_read_b128 = 4
// consecutive k per lane, same k permutation for both operands); the next chunk's global loads
// are issued before the current chunk's MFMAs (register prefetch).  16-byte global loads when the
// leading dimensions allow it, scalar otherwise (only the 23-column VFE input).
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int BM = 128, BK = 32, PITCH = BK + 4;

template <bool WT, int NT, bool VEC>
__global__ __launch_bounds__(256, 2) void linear_fwd(const float *__restrict__ x, int ldx, const float *__restrict__ w,
                                                  int ldw, const float *__restrict__ bias, float *__restrict__ y,
                                                  int ldy, double *__restrict__ stats, const float *__restrict__ row_w,
                                                  long long R, int K, int N, int relu, int k_per_split,
                                                  unsigned *__restrict__ done_counter, double fin_eps,
                                                  float *__restrict__ fin_mean_inv, FrameMap fm) {
    constexpr int BNL = 32 * NT;
    constexpr int XV = BM * BK / 4 / 256;          // float4 per thread for the x tile (4)
    constexpr int WV = BNL * BK / 4 / 256;         // float4 per thread for the w tile (NT)
    __shared__ __attribute__((aligned(16))) float s_x[BM * PITCH];
    __shared__ __attribute__((aligned(16))) float s_w[BNL * PITCH];
    __shared__ double s_red[4][2 * BNL];
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
        __syncthreads();
        store_tiles(k0);
        __syncthreads();
        if (k0 + BK < kend) load_tiles(k0 + BK);
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

    // Epilogue.  Every load it needs (bias, row weights) is issued FIRST and unconditionally from clamped addresses, the
    // accumulators become the outputs in place, and only then come the stores: vector-memory operations return in order
    // (one vmcnt), so a load issued between stores -- or under a condition the waitcnt pass cannot see through -- made
    // every store wait for all earlier ones (`s_waitcnt vmcnt(0)` in front of each of the 16 * NT stores).
    float bsv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int c = n0 + t * 32 + li;
        bsv[t] = bias ? bias[c < N ? c : N - 1] : 0.f;
    }
    float rwv[16];
    if (stats && row_w) {                      // one uniform branch around the whole batch of loads
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const long long gr = r0 + wv * 32 + row;
            rwv[r] = row_w[gr < R ? gr : R - 1];
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) rwv[r] = 1.f;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = acc[t][r] + bsv[t];
            if (relu) v = fmaxf(v, 0.f);
            acc[t][r] = v;
        }
    // store, then the BatchNorm sums.  A 128-row block almost always lies inside one frame; a block that straddles a
    // frame boundary repeats the (register-only) reduction once per frame with the other frames' rows masked out.
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int c = n0 + t * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const long long gr = r0 + wv * 32 + row;
            if (gr < R && c < N) y[gr * ldy + c] = acc[t][r];
        }
    }
    if (stats) {
        const long long r_last = (r0 + BM - 1 < R ? r0 + BM - 1 : R - 1);
        const int f_lo = fm_frame_of(fm, r0), f_hi = fm.F == 1 ? 0 : fm_frame_of(fm, r_last);
        const int s_lo = fm.F == 1 ? 0 : fm_seg_of(fm, r0), s_hi = fm.F == 1 ? 0 : fm_seg_of(fm, r_last);
        // frames met by this block, in segment order (a block can cross from the real rows into the padded rows, whose
        // frame order starts again at 0): walk the segments, one reduction per segment
        for (int sg = s_lo; sg <= s_hi; ++sg) {
            const int f = fm.F == 1 ? 0 : (int)fm.seg_frame[sg];
            const long long lo = fm.F == 1 ? 0 : fm.bound[sg], hi = fm.F == 1 ? R : fm.bound[sg + 1];
            double s1[NT], s2[NT];          // f64 from the first addition on (var = E[y^2] - mean^2 cancels)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int c = n0 + t * 32 + li;
                s1[t] = 0.0; s2[t] = 0.0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const long long gr = r0 + wv * 32 + row;
                    float v = acc[t][r];
                    asm volatile("" : "+v"(v));     // opaque per segment: keeps the 64 f64 conversions and squares from being
                                                    // hoisted out of the segment loop (they cost 256 VGPRs = the second wave)
                    if (gr < R && c < N && gr >= lo && gr < hi) {
                        const double rw = (double)rwv[r];
                        s1[t] += rw * (double)v;
                        s2[t] += rw * (double)v * (double)v;
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const double a = s1[t] + __shfl_xor(s1[t], 32, 64), b = s2[t] + __shfl_xor(s2[t], 32, 64);
                if (lh == 0) { s_red[wv][t * 32 + li] = a; s_red[wv][BNL + t * 32 + li] = b; }
            }
            __syncthreads();
            double *fstats = stats + (size_t)f * MVX_REP * 2 * N;
            for (int e = tid; e < 2 * BNL; e += 256) {
                const int which = e / BNL, c = e % BNL;
                if (n0 + c < N) {
                    const double t = s_red[0][e] + s_red[1][e] + s_red[2][e] + s_red[3][e];
                    atomicAdd(fstats + ((size_t)(blockIdx.y % MVX_REP) * 2 + which) * N + n0 + c, t);
                }
            }
        }
        (void)f_lo; (void)f_hi;
        if (done_counter) {
            __shared__ int s_last;
            bn_finalize_by_last_block(done_counter, gridDim.x * gridDim.y, stats, N, fm, fin_eps, fin_mean_inv, &s_last);
        }
    }
}

// dW partial: slab[strip][n][k] over the rows of the strip.  Workgroup block = 128(n) x 128(k),
// wave (wn, wk) owns 64 x 64 = 2 x 2 MFMA tiles; the reduction runs over rows, 32 per LDS step.
constexpr int WR = 32;
template <bool VEC>
__global__ __launch_bounds__(256) void linear_wgrad(const float *__restrict__ x, int ldx, const float *__restrict__ dz,
                                                    int lddz, float *__restrict__ slabs, long long R, int K, int N,
                                                    long long rows_per_strip) {
    __shared__ __attribute__((aligned(16))) float s_z[WR * 128];
    __shared__ __attribute__((aligned(16))) float s_x[WR * 128];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int wn = wv >> 1, wk = wv & 1;
    const int n0 = blockIdx.y * 128, k0 = blockIdx.z * 128;
    const long long rbeg = (long long)blockIdx.x * rows_per_strip;
    const long long rend = rbeg + rows_per_strip < R ? rbeg + rows_per_strip : R;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const bool wave_on = (n0 + wn * 64 < N) && (k0 + wk * 64 < K);

    float4 zr[4], xr[4];
    auto load_tiles = [&](long long rr) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = tid + 256 * u, r = c >> 5, part = c & 31;
            const long long gr = rr + r;
            const bool ok = gr < rend;
            if (VEC) {
                zr[u] = (ok && n0 + part * 4 < N) ? *(const float4 *)(dz + gr * lddz + n0 + part * 4) : make_float4(0, 0, 0, 0);
                xr[u] = (ok && k0 + part * 4 < K) ? *(const float4 *)(x + gr * ldx + k0 + part * 4) : make_float4(0, 0, 0, 0);
            } else {
                float zz[4], xx[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    zz[j] = (ok && n0 + part * 4 + j < N) ? dz[gr * lddz + n0 + part * 4 + j] : 0.f;
                    xx[j] = (ok && k0 + part * 4 + j < K) ? x[gr * ldx + k0 + part * 4 + j] : 0.f;
                }
                zr[u] = make_float4(zz[0], zz[1], zz[2], zz[3]);
                xr[u] = make_float4(xx[0], xx[1], xx[2], xx[3]);
            }
        }
    };
    load_tiles(rbeg);
    for (long long rr = rbeg; rr < rend; rr += WR) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = tid + 256 * u;
            *(float4 *)(s_z + c * 4) = zr[u];
            *(float4 *)(s_x + c * 4) = xr[u];
        }
        __syncthreads();
        if (rr + WR < rend) load_tiles(rr + WR);
        if (wave_on) {
#pragma unroll 4
            for (int kk = 0; kk < WR / 2; ++kk) {
                const int row = 2 * kk + lh;
                const float a0 = s_z[row * 128 + wn * 64 + li], a1 = s_z[row * 128 + wn * 64 + 32 + li];
                const float b0 = s_x[row * 128 + wk * 64 + li], b1 = s_x[row * 128 + wk * 64 + 32 + li];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
    }
    if (wave_on) {
        float *o = slabs + (size_t)blockIdx.x * N * K;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = n0 + wn * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const int k = k0 + wk * 64 + b * 32 + li;
                    if (n < N && k < K) o[(size_t)n * K + k] = acc[a][b][r];
                }
    }
}

// out[e] = sum over slabs, in a FIXED order: lane l of a 16-lane group adds slabs l, l+16, ... sequentially, the 16
// partials are combined by a butterfly (same tree every run).  Sixteen loads in flight per element instead of a
// dependent chain of `nslabs` loads: these reductions are small (<= 100 k elements) and were pure latency.
__global__ __launch_bounds__(256) void slab_reduce(const float *__restrict__ slabs, float *__restrict__ out, size_t per, int nslabs,
                                                   int accumulate) {
    const int sub = threadIdx.x & 15;
    for (size_t e = blockIdx.x * (size_t)16 + (threadIdx.x >> 4); e < per; e += (size_t)gridDim.x * 16) {
        float s = 0.f;
        for (int k = sub; k < nslabs; k += 16) s += slabs[(size_t)k * per + e];
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) s += __shfl_xor(s, d, 16);
        if (sub == 0) out[e] = accumulate ? out[e] + s : s;
    }
}

// large outputs (split-K of a tall product): one thread per element, coalesced, few slabs
__global__ void slab_reduce_wide(const float *__restrict__ slabs, float *__restrict__ out, size_t per, int nslabs, int accumulate) {
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < per; e += (size_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < nslabs; ++k) s += slabs[(size_t)k * per + e];
        out[e] = accumulate ? out[e] + s : s;
    }
}

inline void launch_slab_reduce(const float *slabs, float *out, size_t total, int nslabs, int accumulate, hipStream_t st) {
    if (total > (1u << 18))
        hipLaunchKernelGGL(slab_reduce_wide, dim3(mvx_cdiv(total, 256) > 2048 ? 2048 : mvx_cdiv(total, 256)), dim3(256), 0, st,
                           slabs, out, total, nslabs, accumulate);
    else
        hipLaunchKernelGGL(slab_reduce, dim3(mvx_cdiv(total, 16) > 8192 ? 8192 : mvx_cdiv(total, 16)), dim3(256), 0, st, slabs,
                           out, total, nslabs, accumulate);
}

inline long long strip_rows(long long R, int N, int K) {
    const long long blocks = (long long)mvx_cdiv(N, 128) * mvx_cdiv(K, 128);
    long long strips = 1536 / blocks;            // enough workgroups to fill 256 CUs a few times over
    // ... without making the slab sum long: 96 strips for the wide layers; the narrow ones (one or two 128 x 128 blocks:
    // the VFE / fusion tail layers, HBM bound on reading x and dz) need more workgroups than 96 to draw bandwidth, and
    // their slabs are small (64 KB per strip and block)
    const long long cap = blocks <= 2 ? 512 / blocks : 96;
    if (strips > cap) strips = cap;
    if (strips < 1) strips = 1;
    long long rows = (R + strips - 1) / strips;
    if (rows < 4 * WR) rows = 4 * WR;
    return ((rows + WR - 1) / WR) * WR;
}

inline bool aligned16(const void *p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace

extern "C" size_t mvx_linear_splitk_workspace_bytes(int64_t rows, int32_t n) {
    return rows > 0 && n > 0 ? (size_t)16 * rows * n * sizeof(float) : 0;
}

static int linear_forward_impl(const float *x, int32_t ldx, const float *w, int32_t ldw, int32_t w_transposed,
                               const float *bias, float *y, int32_t ldy, double *stats, const float *row_w,
                               int64_t rows, int32_t k, int32_t n, int32_t flags, void *splitk_workspace,
                               size_t splitk_workspace_bytes, unsigned *fin_counter, double fin_count, double fin_eps,
                               float *fin_mean_inv, const mvx_frames_t *frames, int row_kind, void *stream) {
    SplitAmax am = mvxi_take_split_amax();               // x bound for this call (fp16 pieces); cleared whatever kernel runs
    am.coarse_a = (flags & MVX_FLAG_AMAX_COARSE) ? 1 : 0;
    const int relu = flags & MVX_FLAG_RELU;
    MVX_CHECK_ARG(x && w && y && rows >= 0 && k > 0 && n > 0 && ldx >= k && ldy >= n);
    MVX_CHECK_ARG(ldw >= (w_transposed ? n : k));
    hipStream_t st = (hipStream_t)stream;
    FrameMap fm;
    MVX_CHECK_ARG(mvx_build_frame_map(fm, frames, row_kind, rows, fin_count));
    if (stats && !(flags & MVX_FLAG_PREZEROED)) {
        hipError_t e = hipMemsetAsync(stats, 0, sizeof(double) * MVX_REP * 2 * n * fm.F, st);
        if (e != hipSuccess) return (int)e;
    }
    if (rows == 0) return MVX_OK;
    // 16-byte loads need: aligned bases, leading dimensions and (for chunk tails) K, N multiples of 4
    const bool vec = aligned16(x) && aligned16(w) && ldx % 4 == 0 && ldw % 4 == 0 && k % 4 == 0 &&
                     (!w_transposed || n % 4 == 0);
    const bool wide = n > 64;
    // Skinny problems (few row/column blocks, long K) without an epilogue are split along K into
    // slabs that a second kernel sums in a fixed order (deterministic); needs a workspace.
    int splits = 1;
    const long long blocks = (long long)mvx_cdiv(rows, BM) * mvx_cdiv(n, wide ? 128 : 64);
    if (splitk_workspace && !bias && !stats && !relu && ldy == n && blocks < 128 && k >= 8 * BK) {
        splits = (int)(256 / blocks);
        if (splits > k / (2 * BK)) splits = k / (2 * BK);
        if (splits > 16) splits = 16;
        if (splits < 1) splits = 1;
    }
    int k_per_split = ((mvx_cdiv(k, splits) + BK - 1) / BK) * BK;
    splits = (int)mvx_cdiv(k, k_per_split);
    float *ydst = y;
    int ld_dst = ldy;
    if (splits > 1) {
        MVX_CHECK_ARG(splitk_workspace_bytes >= (size_t)splits * rows * n * sizeof(float));
        ydst = (float *)splitk_workspace;
        ld_dst = n;
    }
    if ((flags & MVX_FLAG_SPLIT) && vec && wide && splits == 1 && !w_transposed && mvxi_rowgemm_k128_ok(ldx, ldw, ldy, k, n))
        return mvxi_linear_forward_k128(x, ldx, w, ldw, bias, y, ldy, stats, row_w, (long long)rows, n, relu, fin_counter, fin_eps,
                                        fin_mean_inv, fm, mvx_split_code(flags), st, am);
    if ((flags & MVX_FLAG_SPLIT) && vec && wide && splits == 1 && !w_transposed)    // bf16x3 arithmetic for the wide layers
        return mvxi_linear_forward_split(x, ldx, w, ldw, bias, y, ldy, stats, row_w, (long long)rows, k, n, relu, fin_counter,
                                         fin_eps, fin_mean_inv, fm, mvx_split_code(flags), st, am);
    const dim3 grid(mvx_cdiv(n, wide ? 128 : 64), mvx_cdiv(rows, BM), splits);
#define MVX_LAUNCH_LIN(WT, NT, VEC)                                                                               \
    hipLaunchKernelGGL((linear_fwd<WT, NT, VEC>), grid, dim3(256), 0, st, x, ldx, w, ldw, bias, ydst, ld_dst, stats, \
                       row_w, (long long)rows, k, n, relu, k_per_split, fin_counter, fin_eps, fin_mean_inv, fm)
    if (w_transposed) {
        if (wide) { if (vec) MVX_LAUNCH_LIN(true, 4, true); else MVX_LAUNCH_LIN(true, 4, false); }
        else      { if (vec) MVX_LAUNCH_LIN(true, 2, true); else MVX_LAUNCH_LIN(true, 2, false); }
    } else {
        if (wide) { if (vec) MVX_LAUNCH_LIN(false, 4, true); else MVX_LAUNCH_LIN(false, 4, false); }
        else      { if (vec) MVX_LAUNCH_LIN(false, 2, true); else MVX_LAUNCH_LIN(false, 2, false); }
    }
#undef MVX_LAUNCH_LIN
    MVX_LAUNCH_CHECK();
    if (splits > 1) {
        if (ldy == n) {
            const size_t total = (size_t)rows * n;
            launch_slab_reduce((const float *)splitk_workspace, y, total, splits, 0, st);
        } else {
            return MVX_EINVAL;   // split-K needs a dense destination
        }
        MVX_LAUNCH_CHECK();
    }
    return MVX_OK;
}

extern "C" int mvx_linear_forward(const float *x, int32_t ldx, const float *w, int32_t ldw, int32_t w_transposed,
                                  const float *bias, float *y, int32_t ldy, double *stats, const float *row_w,
                                  int64_t rows, int32_t k, int32_t n, int32_t flags, void *splitk_workspace,
                                  size_t splitk_workspace_bytes, void *stream) {
    return linear_forward_impl(x, ldx, w, ldw, w_transposed, bias, y, ldy, stats, row_w, rows, k, n, flags, splitk_workspace,
                               splitk_workspace_bytes, nullptr, 1.0, 0.0, nullptr, nullptr, MVX_ROWS_SINGLE, stream);
}

extern "C" int mvx_linear_forward_bn(const float *x, int32_t ldx, const float *w, int32_t ldw, int32_t w_transposed,
                                     const float *bias, float *y, int32_t ldy, double *stats, const float *row_w,
                                     int64_t rows, int32_t k, int32_t n, int32_t flags, uint32_t *done_counter,
                                     double count, double eps, float *mean_inv, void *stream) {
    MVX_CHECK_ARG(stats && done_counter && mean_inv && count > 0 && rows > 0);
    if (!(flags & MVX_FLAG_PREZEROED)) {
        hipError_t e = hipMemsetAsync(done_counter, 0, sizeof(uint32_t), (hipStream_t)stream);
        if (e != hipSuccess) return (int)e;
    }
    return linear_forward_impl(x, ldx, w, ldw, w_transposed, bias, y, ldy, stats, row_w, rows, k, n, flags, nullptr, 0,
                               done_counter, count, eps, mean_inv, nullptr, MVX_ROWS_SINGLE, stream);
}

extern "C" int mvx_linear_forward_bn_frames(const float *x, int32_t ldx, const float *w, int32_t ldw, int32_t w_transposed,
                                            const float *bias, float *y, int32_t ldy, double *stats, const float *row_w,
                                            int64_t rows, int32_t k, int32_t n, int32_t flags, uint32_t *done_counter,
                                            double eps, float *mean_inv, const mvx_frames_t *frames_host, int32_t row_kind,
                                            void *stream) {
    MVX_CHECK_ARG(stats && done_counter && mean_inv && rows > 0 && frames_host);
    if (!(flags & MVX_FLAG_PREZEROED)) {
        hipError_t e = hipMemsetAsync(done_counter, 0, sizeof(uint32_t), (hipStream_t)stream);
        if (e != hipSuccess) return (int)e;
    }
    return linear_forward_impl(x, ldx, w, ldw, w_transposed, bias, y, ldy, stats, row_w, rows, k, n, flags, nullptr, 0,
                               done_counter, 1.0, eps, mean_inv, frames_host, row_kind, stream);
}

extern "C" size_t mvx_linear_wgrad_workspace_bytes(int64_t rows, int32_t k, int32_t n) {
    if (rows <= 0 || k <= 0 || n <= 0) return 256;
    const long long per = strip_rows(rows, n, k);
    const long long strips = (rows + per - 1) / per;
    return (size_t)strips * n * k * sizeof(float);
}

extern "C" int mvx_linear_wgrad(const float *x, int32_t ldx, const float *dz, int32_t lddz, float *dw,
                                int64_t rows, int32_t k, int32_t n, int32_t flags, void *workspace,
                                size_t workspace_bytes, void *stream) {
    const SplitAmax am = mvxi_take_split_amax();         // (x, dz) bound for this call (fp16 pieces)
    MVX_CHECK_ARG(x && dz && dw && workspace && rows >= 0 && k > 0 && n > 0 && ldx >= k && lddz >= n);
    hipStream_t st = (hipStream_t)stream;
    if (rows == 0) {
        if (flags & MVX_FLAG_ACCUMULATE) return MVX_OK;
        hipError_t e = hipMemsetAsync(dw, 0, sizeof(float) * (size_t)n * k, st);
        return e == hipSuccess ? MVX_OK : (int)e;
    }
    const long long per = strip_rows(rows, n, k);
    const long long strips = (rows + per - 1) / per;
    MVX_CHECK_ARG(workspace_bytes >= (size_t)strips * n * k * sizeof(float));
    const bool vec = aligned16(x) && aligned16(dz) && ldx % 4 == 0 && lddz % 4 == 0 && k % 4 == 0 && n % 4 == 0;
    const dim3 grid((unsigned)strips, mvx_cdiv(n, 128), mvx_cdiv(k, 128));
    if (vec && (flags & MVX_FLAG_SPLIT)) {
        // rows_per_strip is a multiple of the 32-row LDS step in both kernels (strip_rows)
        int rc = mvxi_linear_wgrad_split(x, ldx, dz, lddz, (float *)workspace, (long long)rows, k, n, per, strips,
                                         mvx_split_code(flags), st, am);
        if (rc) return rc;
    } else if (vec)
        hipLaunchKernelGGL(linear_wgrad<true>, grid, dim3(256), 0, st, x, ldx, dz, lddz, (float *)workspace,
                           (long long)rows, k, n, per);
    else
        hipLaunchKernelGGL(linear_wgrad<false>, grid, dim3(256), 0, st, x, ldx, dz, lddz, (float *)workspace,
                           (long long)rows, k, n, per);
    MVX_LAUNCH_CHECK();
    const size_t total = (size_t)n * k;
    launch_slab_reduce((const float *)workspace, dw, total, (int)strips, flags & MVX_FLAG_ACCUMULATE, st);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}
