// Row GEMMs with a SHALLOW reduction (K = 128) in split arithmetic: weights stay, rows stream.
//
// y[r][n] = [ReLU](sum_k x[r][k] W[n][k] + b[n]) for the layers whose input is 128 wide -- the head FCN of the VFE stack and its
// input gradient (modules/voxelnet/VoxelNet.py:28-33: Linear(128, 128) over every point row), the 128 -> 128 layer of the fusion
// MLP and the input gradient of its 768 -> 128 layer (modules/imhead/Pipe.py:94-104) -- the same contract as linear_fwd_split
// (linear_split.hip: per-frame BatchNorm sums, in-kernel finalisation, fp16 ranges) and the SAME numbers: the products of a 16-k
// step are issued in the same order and the steps follow each other in the same order, so y is bit-identical to that kernel's.
//
// Why another kernel.  With K = 128 a 128 x 128 tile of linear_fwd_split does four 32-deep chunks of matrix work (192 MFMAs per
// wave) around which it loads, cuts and stages 64 KB of x AND 64 KB of weights, meets at eight barriers and reduces its BatchNorm
// sums: 205 us for the 320 k x 128 x 128 FCN of BASELINE config 2 where the MFMAs need 50 and HBM 41 (profiles/r05_k128_*).
// Here a workgroup (eight waves, one per CU) cuts its 128 columns of W ONCE into LDS (NP planes, 272-byte rows) and then every
// wave walks 32-row blocks of x on its own: the block's operand goes from global memory straight into the registers the MFMA
// reads it from (lane (row, k half) loads the 32 bytes of its 8 k values per step: no LDS, no barrier), is cut there, and meets
// the weight fragments from LDS.  Waves never wait for each other, so their loads, cuts and MFMAs interleave on a SIMD by
// themselves.  The next block's operand is requested half a block ahead.  N > 128: column chunk c of the output belongs to the
// workgroups {c, c + chunks, ...}; the rows are then read once per chunk (they are L2 / MALL resident: 128 floats per row).
#include "common.h"
#include "split_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int K128 = 128, NCH = 128, WROW = K128 * 2 + 16;       // weight row in LDS: 128 pieces + 16 B pad (17 16-byte slots)

template <int NP, int FMT>
__global__ __launch_bounds__(512) void rowgemm_k128(const float *__restrict__ x, int ldx, const float *__restrict__ w, int ldw,
                                                    const float *__restrict__ bias, float *__restrict__ y, int ldy,
                                                    double *__restrict__ stats, const float *__restrict__ row_w, long long R,
                                                    int N, int relu, unsigned *__restrict__ done_counter, double fin_eps,
                                                    float *__restrict__ fin_mean_inv, FrameMap fm,
                                                    const float *__restrict__ x_amax, int x_coarse, int chunks) {
    __shared__ __attribute__((aligned(16))) unsigned char s_w[NP * NCH * WROW];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 31, lh = lane >> 5;
    const int chunk = blockIdx.x % chunks, wg = blockIdx.x / chunks, wgs = (gridDim.x - chunk + chunks - 1) / chunks;
    const int n0 = chunk * NCH;
    const int ntl = N - n0 >= NCH ? 4 : (N - n0) / 32;      // 32-column tiles of this chunk (the last chunk of an N that is not a multiple of 128)
    float x_scale = 1.f;
    if constexpr (FMT == 1) x_scale = x_coarse ? split_scale_coarse(x_amax) : split_scale_of(x_amax);

    // ---- the chunk's weights, cut once: [piece][n][k] rows of 272 bytes
    for (int c = tid; c < NCH * K128 / 4; c += 512) {
        const int n = c / (K128 / 4), part = c % (K128 / 4);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (n0 + n < N) v = *(const f32x4 *)(w + (long long)(n0 + n) * ldw + part * 4);
        if constexpr (FMT == 1) v *= SPLIT_F16_WSCALE;
        uint2 pc[NP];
        split_n<NP, FMT>(v[0], v[1], v[2], v[3], pc);
#pragma unroll
        for (int q = 0; q < NP; ++q) *(uint2 *)(s_w + (q * NCH + n) * WROW + part * 8) = pc[q];
    }
    __syncthreads();

    const long long nblocks = (R + 31) / 32;
    const long long stride = (long long)wgs * 8;
    long long b = (long long)wg * 8 + wv;
    const int b_base = li * WROW + lh * 16;
    float bsv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) bsv[t] = bias && t < ntl ? bias[n0 + t * 32 + li] : 0.f;

    // operand registers of one block: half h2 holds the steps 4 h2 .. 4 h2 + 3 (two float4 = 8 k values per step)
    f32x4 a[2][4][2];
    auto load_half = [&](long long blk, int h2) __attribute__((always_inline)) {
        long long gr = blk * 32 + li;
        gr = gr < R ? gr : R - 1;                       // clamped: in-bounds reads, the rows are masked at the stores / sums
        const float *p = x + gr * ldx + h2 * 64 + lh * 8;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            a[h2][s][0] = *(const f32x4 *)(p + s * 16);
            a[h2][s][1] = *(const f32x4 *)(p + s * 16 + 4);
        }
    };
    f32x16 acc[4];
    auto mac_half = [&](int h2) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f32x4 v0 = a[h2][s][0], v1 = a[h2][s][1];
            if constexpr (FMT == 1) { v0 *= x_scale; v1 *= x_scale; }
            uint2 p0[NP], p1[NP];
            split_n<NP, FMT>(v0[0], v0[1], v0[2], v0[3], p0);
            split_n<NP, FMT>(v1[0], v1[1], v1[2], v1[3], p1);
            bf16x8 av[NP];
#pragma unroll
            for (int q = 0; q < NP; ++q) av[q] = __builtin_bit_cast(bf16x8, make_uint4(p0[q].x, p0[q].y, p1[q].x, p1[q].y));
            const int ko = (h2 * 4 + s) * 32;
#pragma unroll
            for (int t = 0; t < 4; t += 2) {
                if (t >= ntl) break;                     // block-uniform
                bf16x8 b0[NP], b1[NP];
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    b0[q] = __builtin_bit_cast(bf16x8, *(const uint4 *)(s_w + (q * NCH + t * 32) * WROW + b_base + ko));
                    b1[q] = __builtin_bit_cast(bf16x8, *(const uint4 *)(s_w + (q * NCH + (t + 1) * 32) * WROW + b_base + ko));
                }
                split_mac2<NP, FMT>(acc[t], acc[t + 1], av, b0, b1);
            }
            __builtin_amdgcn_sched_barrier(0);          // one step's cut and fragments at a time (register budget)
        }
    };

    // running per-frame BatchNorm sums of this wave (columns n0 + 32 t + li; the two lane halves hold different rows)
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
    int cur_seg = -1;
    const unsigned rep = (unsigned)((wg * 8 + wv) % MVX_REP);
    auto flush = [&]() __attribute__((always_inline)) {
        if (cur_seg < 0) return;
        const int f = fm.F == 1 ? 0 : (int)fm.seg_frame[cur_seg];
        double *fstats = stats + (size_t)f * MVX_REP * 2 * N + (size_t)rep * 2 * N;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const double u = s1[t] + __shfl_xor(s1[t], 32, 64), v = s2[t] + __shfl_xor(s2[t], 32, 64);
            if (lh == 0 && t < ntl) {
                atomicAdd(fstats + n0 + t * 32 + li, u);
                atomicAdd(fstats + N + n0 + t * 32 + li, v);
            }
            s1[t] = 0.0; s2[t] = 0.0;
        }
        cur_seg = -1;
    };

    if (b < nblocks) { load_half(b, 0); load_half(b, 1); }
    while (b < nblocks) {
        const long long nb = b + stride < nblocks ? b + stride : b;      // unconditional prefetch: the last one re-reads and drops
        float rwl = 1.f;
        if (stats && row_w) { const long long gr = b * 32 + li; rwl = row_w[gr < R ? gr : R - 1]; }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        mac_half(0);
        load_half(nb, 0);
        __builtin_amdgcn_sched_barrier(0);
        mac_half(1);
        load_half(nb, 1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (FMT == 1) {
            const float o_scale = split_inverse(x_scale) * (1.f / SPLIT_F16_WSCALE);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] *= o_scale;
        }
        // ---- epilogue of the block: bias, ReLU, stores (two 128-byte row pieces per instruction), sums
        const long long r0 = b * 32;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[t][r] + bsv[t];
                if (relu) v = fmaxf(v, 0.f);
                acc[t][r] = v;
            }
        // rows of the block as 32-bit offsets from one base pointer (64-bit row numbers per accumulator register cost 32 VGPRs)
        const int live = R - r0 < 32 ? (int)(R - r0) : 32;              // wave-uniform
        float *yb = y + (r0 + 4 * lh) * ldy + n0 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rc = (r & 3) + 8 * (r >> 2);
            if (live == 32 || rc + 4 * lh < live) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (t < ntl) yb[rc * ldy + t * 32] = acc[t][r];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (stats) {
            const long long r_last = r0 + live - 1;
            const int s_lo = fm.F == 1 ? 0 : fm_seg_of(fm, r0), s_hi = fm.F == 1 ? 0 : fm_seg_of(fm, r_last);
            if (s_lo != cur_seg || s_hi != s_lo) flush();
            for (int sg = s_lo; sg <= s_hi; ++sg) {
                const long long lo = fm.F == 1 ? 0 : fm.bound[sg], hi = fm.F == 1 ? R : fm.bound[sg + 1];
                const int lo_rel = lo > r0 ? (int)(lo - r0) : 0, hi_rel = hi - r0 < live ? (int)(hi - r0) : live;   // wave-uniform
                cur_seg = sg;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rc = (r & 3) + 8 * (r >> 2), row = rc + 4 * lh;
                    // the weight of row rc / rc + 4 by lane half: two constant-lane reads instead of a shuffle (whose 16 lane
                    // addresses would live in registers across the whole loop)
                    const float w0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rwl), rc));
                    const float w1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rwl), rc + 4));
                    const double rw = (double)(lh ? w1 : w0);
                    if (row >= lo_rel && row < hi_rel) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const double v = (double)acc[t][r];
                            s1[t] += rw * v;
                            s2[t] += rw * v * v;
                        }
                    }
                }
                if (sg < s_hi) flush();
            }
        }
        b += stride;
    }
    if (stats) {
        flush();
        if (done_counter) {
            __shared__ int s_last;
            bn_finalize_by_last_block(done_counter, gridDim.x, stats, N, fm, fin_eps, fin_mean_inv, &s_last);
        }
    }
}

}  // namespace

// mvx_tuning_set(MVX_TUNE_ROWGEMM_K128, 0 / 1): A/B switch (1 = on, the default)
static int g_k128_on = 1;
void mvxi_rowgemm_k128_enable(long long v) { g_k128_on = v != 0; }

bool mvxi_rowgemm_k128_ok(int ldx, int ldw, int ldy, int k, int n) {
    return g_k128_on && k == K128 && n % 64 == 0 && n >= NCH && ldx % 4 == 0 && ldw % 4 == 0 && ldy >= n;
}

// Launched by linear.hip (linear_forward_impl) for K = 128 when MVX_FLAG_SPLIT is set; same arguments as mvxi_linear_forward_split
int mvxi_linear_forward_k128(const float *x, int ldx, const float *w, int ldw, const float *bias, float *y, int ldy, double *stats,
                             const float *row_w, long long rows, int n, int relu, unsigned *fin_counter, double fin_eps,
                             float *fin_mean_inv, const FrameMap &fm, int pieces, hipStream_t st, const SplitAmax &am) {
    const int chunks = (n + NCH - 1) / NCH;                 // the last one may hold 64 columns
    // one workgroup per CU in all; never more waves than 32-row blocks
    const long long blocks = (rows + 31) / 32;
    long long per_chunk = 256 / chunks;
    if (per_chunk * 8 > blocks) per_chunk = (blocks + 7) / 8;
    if (per_chunk < 1) per_chunk = 1;
    const dim3 grid((unsigned)(per_chunk * chunks));
#define MVX_GO(NP_, F_)                                                                                                        \
    hipLaunchKernelGGL((rowgemm_k128<NP_, F_>), grid, dim3(512), 0, st, x, ldx, w, ldw, bias, y, ldy, stats, row_w, rows, n, relu, \
                       fin_counter, fin_eps, fin_mean_inv, fm, am.a, am.coarse_a, chunks)
    if (pieces == 4) MVX_GO(2, 1);
    else if (pieces == 3) MVX_GO(3, 0);
    else MVX_GO(2, 0);
#undef MVX_GO
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}
