// Row GEMMs on the bf16 matrix cores with fp32-grade accuracy ("bf16x3" split), the opt-in arithmetic of
// `convmath: bf16x3` for the wide row layers (the fusion MLP's 768 -> 768 and 768 -> 128 layers and their input
// gradients: nn.Linear / 1x1 Conv2d of modules/layers/Blocks.py:9,14,35,39 as used by modules/imhead/Pipe.py:88-92).
//
// Same contract as linear_fwd (linear.hip): y[r][n] = [ReLU](sum_k x[r][k] * W[n][k] + b[n]) with optional per-frame
// BatchNorm sums / in-kernel finalisation, W row-major [n][k] (the input gradient passes the transposed weight as a
// row-major matrix: modules/frames.py keeps that copy per parameter version).  Every f32 operand is
// split while it is staged into hi = bf16(v), lo = bf16(v - hi) and a product is hi*hi + hi*lo + lo*hi with f32
// accumulation: three v_mfma_f32_32x32x16_bf16 per product (csrc/conv3d_split.hip has the error analysis: ~2e-5 relative).
//
// Tile: 128 rows x 128 columns per workgroup, wave w owns 32 rows x 128 columns (four accumulator tiles); K in chunks
// of 64: LDS rows are 272 bytes = 64 hi (128 B) | 64 lo (128 B) | 16 B pad (17 16-byte slots, odd: the ds_read_b128
// fragment reads of 32 consecutive rows spread over all banks).  Per 16-k step a wave reads 2 A and 8 B fragments for
// 12 MFMAs.  The next chunk's global loads are issued before the current chunk's MFMAs (register prefetch).
#include "common.h"
#include "split_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int BM = 128, BNL = 128, NT = 4;

// Workgroup -> tile mapping that follows the chip: workgroups are dealt round-robin over the 8 XCDs (blocks h and h + 8 share
// one, MI355X_MICROARCH.md), and every XCD has its own L2.  With the plain (column block fastest) order the 6 column blocks
// of a 768-wide layer that share a 128-row block of x land on 6 different XCDs and each L2 fetches those rows for itself:
// 6 x 246 MB of HBM / Infinity-Cache reads for the 80 k x 768 x 768 layer, which made the GEMM memory-bound at 0.3-0.4 of the
// matrix pipe.  Here the launch is one-dimensional: XCD x = h % 8 runs the row blocks {8 j + x} and walks a row block's column
// blocks in consecutive slots, so the blocks that share rows run on the SAME L2 at the same time.  (Speed only: nothing
// depends on the placement.)  Blocks beyond the last row block exit.
struct XcdTile { unsigned rb, cb; bool on; };
__device__ __forceinline__ XcdTile xcd_tile(unsigned h, unsigned nbx, unsigned nby) {
    const unsigned xcd = h & 7u, s = h >> 3;
    XcdTile t;
    t.rb = (s / nbx) * 8u + xcd;
    t.cb = s % nbx;
    t.on = t.rb < nby;
    return t;
}
static inline unsigned xcd_grid(unsigned nbx, unsigned nby) { return 8u * ((nby + 7u) / 8u) * nbx; }

// NP pieces per operand; K in chunks of BK (64 for bf16x3, 32 for bf16x6: 61 KB of LDS either way, two workgroups per CU)
template <int NP, int BK, int FMT>
__global__ __launch_bounds__(256, 2) void linear_fwd_split(const float *__restrict__ x, int ldx, const float *__restrict__ w,
                                                        int ldw, const float *__restrict__ bias, float *__restrict__ y,
                                                        int ldy, double *__restrict__ stats, const float *__restrict__ row_w,
                                                        long long R, int K, int N, int relu,
                                                        unsigned *__restrict__ done_counter, double fin_eps,
                                                        float *__restrict__ fin_mean_inv, FrameMap fm,
                                                        const float *__restrict__ x_amax, int x_coarse,
                                                        unsigned nbx, unsigned nby) {
    const XcdTile tile = xcd_tile(blockIdx.x, nbx, nby);
    if (!tile.on) return;
    // fp16 pieces (FMT = 1, split_common.h): x is scaled by x_scale (from its bound amax, else 1), w by SPLIT_F16_WSCALE; the
    // accumulators are scaled back in front of the epilogue
    float x_scale = 1.f;
    if constexpr (FMT == 1) x_scale = x_coarse ? split_scale_coarse(x_amax) : split_scale_of(x_amax);
    constexpr int ROWB = NP * BK * 2 + 16;         // LDS row: NP pieces of BK bf16 + 16 B pad (an odd number of 16-byte slots)
    constexpr int PQ = BK / 4;                     // float4 per row and chunk
    constexpr int XV = BM * BK / 4 / 256;          // float4 per thread for the x tile
    constexpr int WV = BNL * BK / 4 / 256;         // ... and for the w tile
    __shared__ __attribute__((aligned(16))) unsigned char s_x[BM * ROWB];
    __shared__ __attribute__((aligned(16))) unsigned char s_w[BNL * ROWB];
    __shared__ double s_red[4][2 * BNL];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 31, lh = lane >> 5;
    const long long r0 = (long long)tile.rb * BM;
    const int n0 = tile.cb * BNL;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int a_base = (wv * 32 + li) * ROWB + lh * 16;
    const int b_base = li * ROWB + lh * 16;

    // prefetch registers: unconditional loads from clamped addresses (see linear_fwd); the K tail is zeroed when stored
    f32x4 xr[XV], wr[WV];
    auto load_tiles = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < XV; ++u) {
            const int c = tid + 256 * u, r = c / PQ, part = c % PQ;
            const long long gr = r0 + r;
            const bool ok = gr < R && k0 + part * 4 < K;
            xr[u] = *(const f32x4 *)(ok ? x + gr * ldx + k0 + part * 4 : x);
        }
#pragma unroll
        for (int u = 0; u < WV; ++u) {
            const int c = tid + 256 * u;
            const int n = c / PQ, part = c % PQ;
            const bool ok = n0 + n < N && k0 + part * 4 < K;
            wr[u] = *(const f32x4 *)(ok ? w + (long long)(n0 + n) * ldw + k0 + part * 4 : w);
        }
    };
    auto store_tiles = [&](int k0) __attribute__((always_inline)) {
        const bool tail = k0 + BK > K;
#pragma unroll
        for (int u = 0; u < XV; ++u) {
            const int c = tid + 256 * u, r = c / PQ, part = c % PQ;
            f32x4 v = xr[u];
            if (tail && k0 + part * 4 >= K) v = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (FMT == 1) v *= x_scale;
            uint2 pc[NP];
            split_n<NP, FMT>(v[0], v[1], v[2], v[3], pc);
#pragma unroll
            for (int q = 0; q < NP; ++q) *(uint2 *)(s_x + r * ROWB + q * 2 * BK + part * 8) = pc[q];
        }
#pragma unroll
        for (int u = 0; u < WV; ++u) {
            const int c = tid + 256 * u;
            f32x4 v = wr[u];
            const int n = c / PQ, part = c % PQ;
            if (tail) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (k0 + part * 4 + j >= K) v[j] = 0.f;
            }
            if constexpr (FMT == 1) v *= SPLIT_F16_WSCALE;
            uint2 pc[NP];
            split_n<NP, FMT>(v[0], v[1], v[2], v[3], pc);
#pragma unroll
            for (int q = 0; q < NP; ++q) *(uint2 *)(s_w + n * ROWB + q * 2 * BK + part * 8) = pc[q];
        }
    };

    load_tiles(0);
    for (int k0 = 0; k0 < K; k0 += BK) {
        __syncthreads();
        store_tiles(k0);
        __syncthreads();
        if (k0 + BK < K) load_tiles(k0 + BK);
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            bf16x8 av[NP];
#pragma unroll
            for (int q = 0; q < NP; ++q) av[q] = __builtin_bit_cast(bf16x8, *(const uint4 *)(s_x + a_base + q * 2 * BK + s * 32));
#pragma unroll
            for (int t = 0; t < NT; t += 2) {
                bf16x8 b0[NP], b1[NP];
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    b0[q] = __builtin_bit_cast(bf16x8, *(const uint4 *)(s_w + b_base + t * 32 * ROWB + q * 2 * BK + s * 32));
                    b1[q] = __builtin_bit_cast(bf16x8, *(const uint4 *)(s_w + b_base + (t + 1) * 32 * ROWB + q * 2 * BK + s * 32));
                }
                split_mac2<NP, FMT>(acc[t], acc[t + 1], av, b0, b1);
            }
        }
    }

    if constexpr (FMT == 1) {
        const float o_scale = split_inverse(x_scale) * (1.f / SPLIT_F16_WSCALE);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] *= o_scale;
    }
    // ---- epilogue: the one of linear_fwd (loads first, then the stores, then the per-frame BatchNorm sums in f64)
    float bsv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int c = n0 + t * 32 + li;
        bsv[t] = bias ? bias[c < N ? c : N - 1] : 0.f;
    }
    float rwv[16];
    if (stats && row_w) {
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
        const int s_lo = fm.F == 1 ? 0 : fm_seg_of(fm, r0), s_hi = fm.F == 1 ? 0 : fm_seg_of(fm, r_last);
        for (int sg = s_lo; sg <= s_hi; ++sg) {
            const int f = fm.F == 1 ? 0 : (int)fm.seg_frame[sg];
            const long long lo = fm.F == 1 ? 0 : fm.bound[sg], hi = fm.F == 1 ? R : fm.bound[sg + 1];
            double s1[NT], s2[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int c = n0 + t * 32 + li;
                s1[t] = 0.0; s2[t] = 0.0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const long long gr = r0 + wv * 32 + row;
                    float v = acc[t][r];
                    asm volatile("" : "+v"(v));
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
                    atomicAdd(fstats + ((size_t)(tile.rb % MVX_REP) * 2 + which) * N + n0 + c, t);
                }
            }
        }
        if (done_counter) {
            __shared__ int s_last;
            bn_finalize_by_last_block(done_counter, nbx * nby, stats, N, fm, fin_eps, fin_mean_inv, &s_last);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Weight gradient of a row layer in bf16x3: slab[strip][n][k] = sum over the rows of the strip of dz[r][n] * x[r][k].
// The MFMA reduction index is the ROW, so both operands are needed row-major along k while memory is channel-major:
// 32-row x 128-column tiles of dz and x are split (hi, lo) while they are staged as [column block of 32][row][32] bf16
// (64-byte rows) and fetched with ds_read_b64_tr_b16, the LDS transpose read (the recipe of conv3d_wgrad_split,
// csrc/conv3d_split.hip): a 16-lane group reads 4 rows x 16 columns and every lane receives 4 consecutive rows of its
// column.  Workgroup block 128(n) x 128(k); wave (wn, wk) owns 64 x 64 = 2 x 2 accumulator tiles: per 16-row k-step 8
// operand fragments (16 transpose reads) feed 12 MFMAs.  Same strips / slabs / slab_reduce as the f32 kernel (linear.hip).
// ------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
constexpr int WRS = 32;                 // rows per LDS step (two MFMA k-steps; 64 rows: 232 VGPRs, one wave per SIMD, 0.517 -> 0.585 ms)

__device__ __forceinline__ bf16x8 tr_frag(const unsigned short *row0, const unsigned short *row1) {
    typedef __attribute__((address_space(3))) s16x4 lds4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4 *)row0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4 *)row1);
    s16x8 v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
    v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(bf16x8, v);
}

template <int NP, int FMT>
__global__ __launch_bounds__(256) void linear_wgrad_split(const float *__restrict__ x, int ldx, const float *__restrict__ dz,
                                                          int lddz, float *__restrict__ slabs, long long R, int K, int N,
                                                          long long rows_per_strip, SplitAmax am, unsigned nblk,
                                                          unsigned strips) {
    // XCD-aware order (see xcd_tile): the nblk = (n / 128) x (k / 128) output blocks of one row strip read the same rows of x
    // and dz -- they run in consecutive slots of ONE XCD and share its L2 instead of being spread over four
    const XcdTile tile = xcd_tile(blockIdx.x, nblk, strips);
    if (!tile.on) return;
    const unsigned nby_ = (unsigned)((N + 127) / 128);
    const unsigned strip = tile.rb, by = tile.cb % nby_, bz = tile.cb / nby_;
    float x_scale = 1.f, z_scale = 1.f;                       // fp16 pieces: operands scaled by their bound amax (split_common.h)
    // x is an activation: the coarse scale (a frame set and a single frame then scale it alike); dz: the fine one
    if constexpr (FMT == 1) { x_scale = split_scale_coarse(am.a); z_scale = split_scale_of(am.b); }
    __shared__ __attribute__((aligned(16))) unsigned short s_z[NP][4][WRS][32];      // [piece][32-column block][row][column]
    __shared__ __attribute__((aligned(16))) unsigned short s_x[NP][4][WRS][32];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int wn = wv >> 1, wk = wv & 1;
    const int n0 = by * 128, k0 = bz * 128;
    const long long rbeg = (long long)strip * rows_per_strip;
    const long long rend = rbeg + rows_per_strip < R ? rbeg + rows_per_strip : R;
    // transpose-read roles of this lane (conv3d_wgrad_split)
    const int grp = lane >> 4, i16 = lane & 15, q = i16 >> 2, pcol = (grp & 1) * 16 + 4 * (i16 & 3), kbase = (grp >> 1) * 8;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const bool wave_on = (n0 + wn * 64 < N) && (k0 + wk * 64 < K);

    constexpr int NU = WRS * 32 / 256;       // 16-byte pieces per thread and tile
    f32x4 zr[NU], xr[NU];
    auto load_tiles = [&](long long rr) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int c = tid + 256 * u, r = c >> 5, part = c & 31;
            const long long gr = rr + r;
            const bool ok = gr < rend;
            zr[u] = (ok && n0 + part * 4 < N) ? *(const f32x4 *)(dz + gr * lddz + n0 + part * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            xr[u] = (ok && k0 + part * 4 < K) ? *(const f32x4 *)(x + gr * ldx + k0 + part * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    load_tiles(rbeg);
    for (long long rr = rbeg; rr < rend; rr += WRS) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int c = tid + 256 * u, r = c >> 5, part = c & 31;
            uint2 pc[NP];
            if constexpr (FMT == 1) { zr[u] *= z_scale; xr[u] *= x_scale; }
            split_n<NP, FMT>(zr[u][0], zr[u][1], zr[u][2], zr[u][3], pc);
#pragma unroll
            for (int p = 0; p < NP; ++p) *(uint2 *)(&s_z[p][part >> 3][r][(part & 7) * 4]) = pc[p];
            split_n<NP, FMT>(xr[u][0], xr[u][1], xr[u][2], xr[u][3], pc);
#pragma unroll
            for (int p = 0; p < NP; ++p) *(uint2 *)(&s_x[p][part >> 3][r][(part & 7) * 4]) = pc[p];
        }
        __syncthreads();
        if (rr + WRS < rend) load_tiles(rr + WRS);
        {   // every wave computes (a wave whose 64 x 64 block lies outside n x k multiplies staged zeros and stores nothing): a
            // branch around the MFMAs made the compiler keep the accumulators in VGPRs across the loop and copy all 64 of them into
            // AGPRs and back around the MFMA block of EVERY row step (64 v_accvgpr_write per 48 MFMAs in the ISA)
#pragma unroll
            for (int ks = 0; ks < WRS / 16; ++ks) {
                const int r0 = ks * 16 + kbase + q, r1 = r0 + 4;
                bf16x8 az[2][NP], bx[2][NP];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        az[t][p] = tr_frag(&s_z[p][wn * 2 + t][r0][pcol], &s_z[p][wn * 2 + t][r1][pcol]);
                        bx[t][p] = tr_frag(&s_x[p][wk * 2 + t][r0][pcol], &s_x[p][wk * 2 + t][r1][pcol]);
                    }
#pragma unroll
                for (int a = 0; a < 2; ++a) split_mac2<NP, FMT>(acc[a][0], acc[a][1], az[a], bx[0], bx[1]);
            }
        }
    }
    if (wave_on) {
        float *o = slabs + (size_t)strip * N * K;
        if constexpr (FMT == 1) {
            const float o_scale = split_inverse(x_scale) * split_inverse(z_scale);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] *= o_scale;
        }
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

}  // namespace

// Launched by linear.hip (mvx_linear_wgrad) when MVX_FLAG_SPLIT is set and the operands are 16-byte aligned.
int mvxi_linear_wgrad_split(const float *x, int ldx, const float *dz, int lddz, float *slabs, long long rows, int k, int n,
                            long long rows_per_strip, long long strips, int pieces, hipStream_t st, const SplitAmax &am) {
    const unsigned nblk = mvx_cdiv(n, 128) * mvx_cdiv(k, 128);
    const dim3 grid(xcd_grid(nblk, (unsigned)strips));
    if (pieces == 4)
        hipLaunchKernelGGL((linear_wgrad_split<2, 1>), grid, dim3(256), 0, st, x, ldx, dz, lddz, slabs, rows, k, n, rows_per_strip, am,
                           nblk, (unsigned)strips);
    else if (pieces == 3)
        hipLaunchKernelGGL((linear_wgrad_split<3, 0>), grid, dim3(256), 0, st, x, ldx, dz, lddz, slabs, rows, k, n, rows_per_strip, am,
                           nblk, (unsigned)strips);
    else
        hipLaunchKernelGGL((linear_wgrad_split<2, 0>), grid, dim3(256), 0, st, x, ldx, dz, lddz, slabs, rows, k, n, rows_per_strip, am,
                           nblk, (unsigned)strips);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

// Launched by linear.hip (linear_forward_impl) when MVX_FLAG_SPLIT is set and the shape qualifies.
int mvxi_linear_forward_split(const float *x, int ldx, const float *w, int ldw, const float *bias, float *y,
                              int ldy, double *stats, const float *row_w, long long rows, int k, int n, int relu,
                              unsigned *fin_counter, double fin_eps, float *fin_mean_inv, const FrameMap &fm, int pieces,
                              hipStream_t st, const SplitAmax &am) {
    const unsigned nbx = mvx_cdiv(n, BNL), nby = mvx_cdiv(rows, BM);
    const dim3 grid(xcd_grid(nbx, nby));
#define MVX_GO(NP_, BK_, F_)                                                                                                     \
    hipLaunchKernelGGL((linear_fwd_split<NP_, BK_, F_>), grid, dim3(256), 0, st, x, ldx, w, ldw, bias, y, ldy, stats, row_w, rows, k, \
                       n, relu, fin_counter, fin_eps, fin_mean_inv, fm, am.a, am.coarse_a, nbx, nby)
    if (pieces == 4) MVX_GO(2, 64, 1);
    else if (pieces == 3) MVX_GO(3, 32, 0);
    else MVX_GO(2, 64, 0);
#undef MVX_GO
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}
