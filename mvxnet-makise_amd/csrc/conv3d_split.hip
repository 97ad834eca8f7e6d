// Dense 3x3x3 convolution on the bf16 matrix cores with fp32-grade accuracy ("bf16x3" split).
//
// Every f32 operand x is split on the fly into two bf16 numbers, hi = bf16(x) and
// lo = bf16(x - hi) (together 16 mantissa bits), and a product is evaluated as
//     x*w  ~=  hi_x*hi_w + hi_x*lo_w + lo_x*hi_w            (3 v_mfma_f32_32x32x16_bf16, f32 accumulate)
// The dropped terms are O(2^-16 |x w|): per-product relative error ~2e-5, far inside the 1e-4 bar of
// the fp32 features (tests compare against the float64 oracle).  A bf16 MFMA retires 16x the MACs
// per cycle of the exact-f32 MFMA, so three of them are ~5x faster than one f32 MFMA step.
//
// Structure = conv3d.hip's gather kernel (8x16-site patch x 64 output channels per workgroup, wave w
// owns rows 2w,2w+1; halo staged once per (depth tap, 32-channel chunk)), with:
//   - LDS rows of pieces x 64 B + 16 B pad -> ds_read_b128 operand fetches
//     (8 consecutive k per lane, the native 32x32x16 A/B fragment) are bank-conflict free;
//   - the f32 -> bf16 pieces split of the activations happens while the halo is staged;
//   - with THREE pieces per operand and six MFMAs per product ("bf16x6", MVX_FLAG_SPLIT3) the arithmetic is fp32-grade:
//     hi + mid + lo is the f32 operand exactly and the dropped cross terms are below the rounding of an f32 product;
//   - weights are pre-split by the pack kernel and staged three taps (one kernel row) at a time, so a
//     barrier pair covers 3 taps x 12 MFMAs per wave instead of one tap.
#include "common.h"
#include "split_common.h"

namespace {


constexpr int TH = 8, TW = 16, HH = TH + 2, HW = TW + 2;
constexpr int BK = 32, BN = 64;
constexpr int TH2 = 16;                           // patch height of the 16 x 16-site gather units

struct Geom {
    int Din, Dout, H, W, Cin, Cout, sd, pd, mode;
    int F = 1;                                     // frames stacked along depth: planes [F * Dout] <- [F * Din]
    int tap_lo = 0, tap_hi = 3;                    // in-plane taps (rows AND columns) [tap_lo, tap_hi) carry weight (MVX_FLAG_TAPS2:
                                                   // the 2x2 window of a stride-2 kernel on the space-to-depth image)
    int s2d = 0;                                   // > 0: channels per parity block of that image: the structurally zero (window tap,
                                                   // parity) blocks are not executed (see Geom::s2d in conv3d.hip)
};

// valid window taps of parity block p = pr * 2 + pc as a 4-bit mask, bit (ta * 2 + tb) (conv3d.hip: s2d_tap_mask)
__device__ __forceinline__ unsigned s2d_tap_mask(int p) {
    const int pr = p >> 1, pc = p & 1;
    unsigned m = 8u;
    if (pr) m |= 2u;
    if (pc) m |= 4u;
    if (pr && pc) m |= 1u;
    return m;
}

// source plane of output plane d (a GLOBAL plane index: frame * Dout + plane) for depth tap kd, or -1; planes of
// different frames never connect (same rule as mvx_src_plane / mvx_dst_plane in common.h)
__device__ __forceinline__ int src_depth(const Geom &g, int d, int kd) {
    const int f = d / g.Dout, dl = d - f * g.Dout;
    if (g.mode == 0) {
        const int s = dl * g.sd - g.pd + kd;
        return (s >= 0 && s < g.Din) ? f * g.Din + s : -1;
    }
    const int t = dl + g.pd - kd;
    if (t < 0 || (t % g.sd) != 0) return -1;
    const int s = t / g.sd;
    return s < g.Din ? f * g.Din + s : -1;
}

// torch W[co][ci][kd][kh][kw] -> wsp[kd][tap][32-channel chunk][n][piece][32] (bf16), forward or dgrad view
__global__ void pack_weights_split(const float *__restrict__ w, unsigned short *__restrict__ wsp, int Co, int Ci, int dgrad, int np, int fmt) {
    const int K = dgrad ? Co : Ci, N = dgrad ? Ci : Co;
    const int nch = K / BK;
    const long long total = 27ll * K * N;
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(e % BK);
        long long r = e / BK;
        const int n = (int)(r % N); r /= N;
        const int ch = (int)(r % nch); r /= nch;
        const int tap = (int)(r % 9);
        const int kd = (int)(r / 9);
        const int a = tap / 3, b = tap % 3;
        const int kk = ch * BK + k;
        const int co = dgrad ? kk : n, ci = dgrad ? n : kk;
        const int kh = dgrad ? 2 - a : a, kw = dgrad ? 2 - b : b;
        float x = w[((((long long)co * Ci + ci) * 3 + kd) * 3 + kh) * 3 + kw];
        const long long row = (((long long)kd * 9 + tap) * nch + ch) * N + n;
        for (int q = 0; q < np; ++q) {
            if (fmt == 0) {
                const __bf16 pb = (__bf16)x;
                wsp[(row * np + q) * BK + k] = __builtin_bit_cast(unsigned short, pb);
                x -= (float)pb;
            } else {
                if (q == 0) x *= SPLIT_F16_WSCALE;
                const _Float16 pb = (_Float16)x;
                wsp[(row * np + q) * BK + k] = __builtin_bit_cast(unsigned short, pb);
                x -= (float)pb;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// The gather (forward / dgrad).  Template: NP pieces, BKT input channels per stage (32 or 16), MT site tiles per wave:
//   MT = 1: 8 x 16-site patch per workgroup, wave w owns patch rows 2w, 2w+1 (32 sites) x 64 channels;
//   MT = 2: 16 x 16-site patch, a wave owns 64 sites x 64 channels (four accumulator tiles): half the weight bytes and
//           weight-operand reads per MFMA and twice the matrix work between two barriers.  The activity flags stay per
//           8 x 16 tile (what activity.hip produces): a workgroup covers the tiles (tx, 2 ty) and (tx, 2 ty + 1) and ORs
//           their flags -- computing a background half is exact, just not needed.
// LDS rows hold the NP pieces of a site's (or an output channel's) BKT channels back to back + 16 B pad (an odd number
// of 16-byte slots: the ds_read_b128 fragment reads of 16 consecutive rows cover all banks); the halo row pitch is a
// multiple of 256 B so that the two patch rows a fragment read touches start on the same slot (lane groups
// {0-3,12-15,20-27} / {4-11,16-19,28-31} then hit 16 distinct slots).  Shapes used (launch_gather_split):
//   bf16x3: <2,32,1> 57 KB and <2,32,2> 79 KB;   bf16x6: <3,32,1> 79 KB and <3,16,2> 59 KB -- two workgroups per CU each.
// WIN: the launch has a tap window or structural zeros (MVX_FLAG_TAPS2): rows and columns of the 3 x 3 kernel are skipped by
// block-uniform masks; without it the three kernel rows are unrolled at compile time (8 % faster on the dense launches).
// ------------------------------------------------------------------------------------------
template <int NP, int BKT, int MT, bool WIN, int FMT>
__global__ __launch_bounds__(256, 2) void conv3d_gather_splitT(const float *__restrict__ in,
                                                               const unsigned short *__restrict__ wsp,
                                                               const float *__restrict__ bias,
                                                               float *__restrict__ out, double *__restrict__ stats,
                                                               Geom g, int relu, const int *__restrict__ in_hflag,
                                                               const unsigned char *__restrict__ out_mask,
                                                               const float *__restrict__ bg_pre, int border_active,
                                                               const int *__restrict__ only_tiles,
                                                               unsigned long long *__restrict__ exec_stages,
                                                               const float *__restrict__ in_amax) {
    // fp16 pieces (FMT = 1, split_common.h): `in` is scaled by a_scale (from its bound amax, else 1), the weights were packed
    // times SPLIT_F16_WSCALE; the accumulators are scaled back before the epilogue
    float a_scale = 1.f;
    if constexpr (FMT == 1) a_scale = split_scale_of(in_amax);
    constexpr int THT = TH * MT, HHT = THT + 2;
    constexpr int ROWBT = NP * BKT * 2 + 16;                 // bytes per LDS row
    constexpr int HROW = (HW * ROWBT + 255) / 256 * 256;     // halo row pitch
    constexpr int PPR = NP * BKT / 8;                        // 16-byte pieces per staged weight row
    constexpr int PT = BN * PPR;                             // ... per tap tile
    constexpr int NWU = (3 * PT + 255) / 256;                // ... per thread and kernel row (3 taps)
    constexpr int PARTS = BKT / 4;                           // float4 per site and stage
    constexpr int NH = (HHT * HW * PARTS + 255) / 256;
    constexpr int KS = BKT / 16;                             // MFMA k-steps per stage
    __shared__ __attribute__((aligned(256))) unsigned char s_halo[HHT * HROW];
    __shared__ __attribute__((aligned(16))) unsigned char s_w[3][BN * ROWBT];
    __shared__ float s_red[4][2 * BN];
    const int tiles_x = (g.W + TW - 1) / TW, tiles_y8 = (g.H + TH - 1) / TH;
    const int ntiles8 = tiles_x * tiles_y8;
    const int tx = blockIdx.x % tiles_x, tyb = blockIdx.x / tiles_x;
    const int tx0 = tx * TW, ty0 = tyb * THT;
    const int t_top = (MT * tyb) * tiles_x + tx;
    const bool has_bot = MT == 2 && 2 * tyb + 1 < tiles_y8;
    const int t_bot = has_bot ? t_top + tiles_x : t_top;
    const int d = blockIdx.y, nb = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int nchunks = g.Cin / BKT;
    // background rewrite (see conv3d.hip / activity.hip): restricted launches and constant tiles
    if (only_tiles && !(only_tiles[(size_t)d * ntiles8 + t_top] | only_tiles[(size_t)d * ntiles8 + t_bot])) return;
    // border_active bit 0: border tiles always count as active; bit 1: bg_pre carries the per-depth-tap and position-class
    // constants (see gather_unit in conv3d.hip): interior tiles skip depth taps with a background-only source halo, border
    // tiles without any active source are filled from the class constants
    const bool on_border = tx0 == 0 || ty0 == 0 || tx0 + TW >= g.W || ty0 + THT >= g.H;
    const bool skip_taps = in_hflag && (border_active & 2) && !on_border;
    unsigned skipped = 0;
    int any_flag = 0;
    bool active = true;
    if (in_hflag) {
        for (int kd = 0; kd < 3; ++kd) {
            const int ds = src_depth(g, d, kd);
            if (ds >= 0) {
                const int fl = in_hflag[(size_t)ds * ntiles8 + t_top] | in_hflag[(size_t)ds * ntiles8 + t_bot];
                any_flag |= fl;
                if (skip_taps && !fl) skipped |= 1u << kd;
            }
        }
        active = (((border_active & 1) && on_border) || any_flag) != 0;
    }
    const bool idle_border = in_hflag && (border_active & 2) && on_border && !any_flag;

    f32x16 acc[MT][2];                                   // [site tile m: patch rows 2 MT wv + 2 m, + 1][channel tile: n0, n0 + 32]
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    int a_base[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) a_base[m] = (2 * MT * wv + 2 * m + (li >> 4)) * HROW + (li & 15) * ROWBT + lh * 16;
    const int b_base = li * ROWBT + lh * 16;

    // one kernel row (three tap tiles) of pre-split weights per barrier pair.  NATIVE vector types: an array of HIP's uint4
    // struct stays an alloca, and the backend "promoted" it to LDS (one workgroup per CU)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 wreg[NWU];
    auto load_w3 = [&](int kd, int a, int cc) __attribute__((always_inline)) {
        // global rows are [n][piece][32 channels] per 32-channel chunk; a BKT = 16 stage takes one half of every piece
        const int c32 = (cc * BKT) / BK, half = BKT == 16 ? (cc & 1) : 0;
#pragma unroll
        for (int u = 0; u < NWU; ++u) {
            int c = tid + 256 * u;
            if (3 * PT % 256 && c >= 3 * PT) c = 3 * PT - 1;                 // clamped; not stored
            const int t = c / PT, rem = c - t * PT;
            const int n = rem / PPR, piece = rem - n * PPR;
            const int q = piece / (BKT / 8), sub = piece - q * (BKT / 8);     // bf16 piece of the split, 16-byte part inside it
            const unsigned char *row = (const unsigned char *)wsp +
                (((((size_t)kd * 9 + a * 3 + t) * (g.Cin / BK) + c32) * g.Cout + (size_t)nb * BN + n) * NP + q) * (BK * 2);
            wreg[u] = *(const u32x4 *)(row + half * 32 + sub * 16);
        }
    };
    auto store_w3 = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < NWU; ++u) {
            const int c = tid + 256 * u;
            if (3 * PT % 256 == 0 || c < 3 * PT) {
                const int t = c / PT, rem = c - t * PT;
                const int n = rem / PPR, piece = rem - n * PPR;
                *(u32x4 *)(s_w[t] + n * ROWBT + piece * 16) = wreg[u];
            }
        }
    };

    // Stages = (valid depth tap) x (BKT-channel chunk), software-pipelined like conv3d_gather_pf (conv3d.hip): the halo of
    // stage s + 1 is fetched into registers while stage s computes (split into its pieces when it is written to LDS), the
    // weight rows go to LDS one kernel row at a time with the next row in flight, and the row needed next is always issued
    // BEFORE the long-latency halo fetch (vector-memory returns are in order).
    int kd_l[3] = {0, 0, 0}, ds_l[3] = {0, 0, 0}, nk = 0;
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) {
        const int ds = src_depth(g, d, kd);
        if (ds >= 0 && !((skipped >> kd) & 1u) && !idle_border) {
            if (nk == 0) { kd_l[0] = kd; ds_l[0] = ds; }
            else if (nk == 1) { kd_l[1] = kd; ds_l[1] = ds; }
            else { kd_l[2] = kd; ds_l[2] = ds; }
            ++nk;
        }
    }
    const int nstages = active ? nk * nchunks : 0;
    // executed work in units of the 8 x 16-tile, 32-channel stage (what the roofline prices)
    if (exec_stages && nstages > 0 && threadIdx.x == 0)
        atomicAdd(exec_stages, (unsigned long long)(nk * (g.Cin / BK)) * (has_bot ? 2 : 1));
    auto stage_of = [&](int st, int &kd, int &ds, int &cc) __attribute__((always_inline)) {
        const int i = st / nchunks;
        cc = st - i * nchunks;
        kd = i == 0 ? kd_l[0] : (i == 1 ? kd_l[1] : kd_l[2]);
        ds = i == 0 ? ds_l[0] : (i == 1 ? ds_l[1] : ds_l[2]);
    };
    int h_off[NH], h_lds[NH];     // per-thread halo slots: global float offset inside a (plane, chunk) image or -1; LDS byte offset or -1
#pragma unroll
    for (int u = 0; u < NH; ++u) {
        const int c = tid + 256 * u;
        h_off[u] = -1;
        h_lds[u] = -1;
        if (c < HHT * HW * PARTS) {
            const int r = c / PARTS, part = c - r * PARTS;
            const int hy = r / HW, hx = r - hy * HW;
            const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
            h_lds[u] = hy * HROW + hx * ROWBT + part * 8;
            if (gy >= 0 && gy < g.H && gx >= 0 && gx < g.W) h_off[u] = (gy * g.W + gx) * g.Cin + part * 4;
        }
    }
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 hreg[NH];
    auto load_halo = [&](int st) __attribute__((always_inline)) {
        int kd, ds, cc;
        stage_of(st, kd, ds, cc);
        const float *img = in + (size_t)ds * g.H * g.W * g.Cin + cc * BKT;
#pragma unroll
        for (int u = 0; u < NH; ++u)
            hreg[u] = h_off[u] >= 0 ? *(const f32x4 *)(img + h_off[u]) : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto store_halo = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < NH; ++u)
            if (h_lds[u] >= 0) {
                uint2 pc[NP];
                if constexpr (FMT == 1) hreg[u] *= a_scale;
                split_n<NP, FMT>(hreg[u][0], hreg[u][1], hreg[u][2], hreg[u][3], pc);
#pragma unroll
                for (int q = 0; q < NP; ++q) *(uint2 *)(s_halo + h_lds[u] + q * BKT * 2) = pc[q];
            }
    };
    auto load_wrow = [&](int st, int a) __attribute__((always_inline)) {
        int kd, ds, cc;
        stage_of(st, kd, ds, cc);
        load_w3(kd, a, cc);
    };
    auto compute_row = [&](int a, unsigned colmask) __attribute__((always_inline)) {
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            if (!((colmask >> b) & 1u)) continue;             // block-uniform: taps outside the window / structural zeros
#pragma unroll
            for (int s2 = 0; s2 < KS; ++s2) {
                bf16x8 b0[NP], b1[NP];
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    b0[q] = __builtin_bit_cast(bf16x8, *(const uint4 *)(s_w[b] + b_base + q * BKT * 2 + s2 * 32));
                    b1[q] = __builtin_bit_cast(bf16x8, *(const uint4 *)(s_w[b] + b_base + 32 * ROWBT + q * BKT * 2 + s2 * 32));
                }
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const int a_off = a_base[m] + a * HROW + b * ROWBT + s2 * 32;
                    bf16x8 av[NP];
#pragma unroll
                    for (int q = 0; q < NP; ++q) av[q] = __builtin_bit_cast(bf16x8, *(const uint4 *)(s_halo + a_off + q * BKT * 2));
                    split_mac2<NP, FMT>(acc[m][0], acc[m][1], av, b0, b1);
                }
            }
        }
    };
    if constexpr (!WIN) {
        if (nstages > 0) {
            load_wrow(0, 0);
            load_halo(0);
        }
        for (int st = 0; st < nstages; ++st) {
            const int nxt = st + 1 < nstages ? st + 1 : st;      // unconditional prefetches: the last stage re-fetches and drops
            __syncthreads();                                      // the previous stage's LDS reads are done
            store_halo();
            store_w3();                                           // kernel row 0
            __syncthreads();
            load_wrow(st, 1);                                     // next weight row first ...
            load_halo(nxt);                                       // ... then the long-latency halo of the next stage
            compute_row(0, 7u);
            __syncthreads();
            store_w3();                                           // kernel row 1
            __syncthreads();
            load_wrow(st, 2);
            compute_row(1, 7u);
            __syncthreads();
            store_w3();                                           // kernel row 2
            __syncthreads();
            load_wrow(nxt, 0);
            compute_row(2, 7u);
        }
    } else {
        // taps executed in a stage, bit (a * 3 + b): all nine, or the window [tap_lo, tap_hi)^2, minus the structural zeros of a
        // stride-2 kernel in space-to-depth form (forward: by the parity block of the stage's input channels; dgrad: of this
        // unit's output channels).  Block-uniform, and never empty (tap (1,1) of the window always carries weight).
        auto taps_of = [&](int st) __attribute__((always_inline)) -> unsigned {
            unsigned m4 = 0xfu;
            if (g.s2d > 0) m4 = s2d_tap_mask(g.mode == 0 ? ((st % nchunks) * BKT) / g.s2d : (nb * BN) / g.s2d);
            unsigned m9 = 0u;
    #pragma unroll
            for (int a = 0; a < 3; ++a)
    #pragma unroll
                for (int b = 0; b < 3; ++b) {
                    const int ta = g.mode == 0 ? a : 2 - a, tb = g.mode == 0 ? b : 2 - b;       // window tap of kernel row / column
                    const bool in = a >= g.tap_lo && a < g.tap_hi && b >= g.tap_lo && b < g.tap_hi;
                    const bool on = g.s2d > 0 ? ((m4 >> ((ta & 1) * 2 + (tb & 1))) & 1u) && ta < 2 && tb < 2 : true;
                    if (in && on) m9 |= 1u << (a * 3 + b);
                }
            return m9;
        };
        auto first_row = [&](unsigned m9) __attribute__((always_inline)) { return (m9 & 7u) ? 0 : ((m9 & 0x38u) ? 1 : 2); };
        auto next_row = [&](unsigned m9, int a) __attribute__((always_inline)) {      // next executed kernel row after a, or -1
            for (int r = a + 1; r < 3; ++r)
                if ((m9 >> (3 * r)) & 7u) return r;
            return -1;
        };
        int st = 0;
        unsigned m9 = nstages > 0 ? taps_of(0) : 0x1ffu;
        int a = first_row(m9);
        if (nstages > 0) {
            load_wrow(0, a);
            load_halo(0);
        }
        while (st < nstages) {
            const bool first = a == first_row(m9);
            int a2 = next_row(m9, a), st2 = st;
            unsigned m9n = m9;
            if (a2 < 0) {
                st2 = st + 1;
                m9n = st2 < nstages ? taps_of(st2) : m9;
                a2 = first_row(m9n);
            }
            const int stp = st2 < nstages ? st2 : st;            // unconditional prefetches: the last one re-fetches and drops
            __syncthreads();                                      // the previous row's LDS reads are done
            if (first) store_halo();
            store_w3();
            __syncthreads();
            load_wrow(stp, a2);                                   // next weight row first ...
            if (first) load_halo(st + 1 < nstages ? st + 1 : st); // ... then the long-latency halo of the next stage
            compute_row(a, (m9 >> (3 * a)) & 7u);
            st = st2; a = a2; m9 = m9n;
        }
    }

    if constexpr (FMT == 1) {
        const float o_scale = split_inverse(split_scale_of(in_amax)) * (1.f / SPLIT_F16_WSCALE);
#pragma unroll
        for (int m = 0; m < MT; ++m) { acc[m][0] *= o_scale; acc[m][1] *= o_scale; }
    }
    // ---- epilogue
    const int n0 = nb * BN + li, n1 = n0 + 32;
    const float bias0 = bias ? bias[n0] : 0.f, bias1 = bias ? bias[n1] : 0.f;
    float skip0 = 0.f, skip1 = 0.f;            // constants of the depth taps that were not executed
    if (skipped && active) {
        const float *bg_tap = bg_pre + (size_t)g.Dout * g.F * g.Cout + (size_t)d * 3 * g.Cout;
#pragma unroll
        for (int kd = 0; kd < 3; ++kd)
            if ((skipped >> kd) & 1u) { skip0 += bg_tap[kd * g.Cout + n0]; skip1 += bg_tap[kd * g.Cout + n1]; }
    }
    float bgv0 = (bg_pre ? bg_pre[(size_t)d * g.Cout + n0] : 0.f) + bias0, bgv1 = (bg_pre ? bg_pre[(size_t)d * g.Cout + n1] : 0.f) + bias1;
    if (relu) { bgv0 = fmaxf(bgv0, 0.f); bgv1 = fmaxf(bgv1, 0.f); }
    float s1a = 0.f, s2a = 0.f, s1b = 0.f, s2b = 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int py = ty0 + 2 * MT * wv + 2 * m;       // first of this site tile's two patch rows
        if (idle_border) {                              // the accumulators are still zero: they take the position-class constants
            const float *bg_cls = bg_pre + (size_t)4 * g.Dout * g.F * g.Cout + (size_t)d * 9 * g.Cout;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int gy = py + (row >> 4), gx = tx0 + (row & 15);
                const int q = 3 * (gy == 0 ? 0 : (gy >= g.H - 1 ? 2 : 1)) + (gx == 0 ? 0 : (gx >= g.W - 1 ? 2 : 1));
                acc[m][0][r] = bg_cls[q * g.Cout + n0];
                acc[m][1][r] = bg_cls[q * g.Cout + n1];
            }
        }
        // site-mask bytes fetched up front (see conv3d.hip: a load between the stores made every store pair wait)
        unsigned site_on = active ? 0xffffu : 0u;
        if (out_mask && active) {
            unsigned char mk[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int gy = min(py + (row >> 4), g.H - 1), gx = min(tx0 + (row & 15), g.W - 1);
                mk[r] = out_mask[((size_t)d * g.H + gy) * g.W + gx];
            }
            site_on = 0u;
#pragma unroll
            for (int r = 0; r < 16; ++r) site_on |= (mk[r] ? 1u : 0u) << r;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int gy = py + (row >> 4), gx = tx0 + (row & 15);
            float v0 = (acc[m][0][r] + skip0) + bias0, v1 = (acc[m][1][r] + skip1) + bias1;
            if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
            if (out_mask && !((site_on >> r) & 1u)) { v0 = bgv0; v1 = bgv1; }
            if (gy < g.H && gx < g.W) {
                float *o = out + (((size_t)d * g.H + gy) * g.W + gx) * g.Cout;
                o[n0] = v0;
                o[n1] = v1;
                s1a += v0; s2a += v0 * v0;
                s1b += v1; s2b += v1 * v1;
            }
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
            const double t = (double)s_red[0][tid] + (double)s_red[1][tid] + (double)s_red[2][tid] + (double)s_red[3][tid];
            const int which = tid / BN, c = tid % BN;
            const unsigned rep = (blockIdx.x + blockIdx.y * gridDim.x) % MVX_REP;
            double *fstats = stats + (size_t)(d / g.Dout) * MVX_REP * 2 * g.Cout;       // the plane's frame
            atomicAdd(fstats + ((size_t)rep * 2 + which) * g.Cout + nb * BN + c, t);
        }
    }
}

// 16 x 16-site units when the launch has enough of them to fill two workgroup slots on every CU with a margin, else 8 x 16.
// The threshold is a tuning value (mvx_tuning_set(MVX_TUNE_SPLIT16_MIN_UNITS, v): 0 forces the 16 x 16 kernel, a huge
// value the 8 x 16 one; the tests run both shapes on small inputs that way).
static long long g_split16_min_units = 768;

extern "C" int mvx_tuning_set(int32_t key, int64_t value) {
    if (key == MVX_TUNE_SPLIT16_MIN_UNITS) { g_split16_min_units = value; return MVX_OK; }
    if (key == MVX_TUNE_GATHER_NARROW_MAX_UNITS) { mvxi_gather_narrow_max_units(value); return MVX_OK; }
    if (key == MVX_TUNE_ROWGEMM_K128) { mvxi_rowgemm_k128_enable(value); return MVX_OK; }
    return MVX_EINVAL;
}

static void launch_gather_split(hipStream_t st, int flags, int planes, int nblocks, const float *in, const unsigned short *wsp,
                                const float *bias, float *out, double *stats, const Geom &g, int relu, const int *in_hflag,
                                const unsigned char *out_mask, const float *bg_pre, int border_active, const int *only_tiles,
                                unsigned long long *exec_stages) {
    const int tiles_x = (int)mvx_cdiv(g.W, TW);
    const long long units16 = (long long)tiles_x * mvx_cdiv(g.H, TH2) * planes * nblocks;
    const bool big = units16 >= g_split16_min_units;
    const dim3 grid(tiles_x * mvx_cdiv(g.H, big ? TH2 : TH), planes, nblocks);
    const bool win = g.tap_lo != 0 || g.tap_hi != 3 || g.s2d > 0;
    const int np = (flags & MVX_FLAG_SPLIT_F16) ? 2 : (flags & MVX_FLAG_SPLIT3) ? 3 : 2;
    const SplitAmax am = mvxi_take_split_amax();         // bound by mvx_split_operand_amax for this launch (fp16 pieces), else NULLs
#define MVX_GO(NP_, BK_, MT_, F_)                                                                                                        \
    do {                                                                                                                             \
        if (win)                                                                                                                     \
            hipLaunchKernelGGL((conv3d_gather_splitT<NP_, BK_, MT_, true, F_>), grid, dim3(256), 0, st, in, wsp, bias, out, stats, g,    \
                               relu, in_hflag, out_mask, bg_pre, border_active, only_tiles, exec_stages, am.a);         \
        else                                                                                                                         \
            hipLaunchKernelGGL((conv3d_gather_splitT<NP_, BK_, MT_, false, F_>), grid, dim3(256), 0, st, in, wsp, bias, out, stats, g,   \
                               relu, in_hflag, out_mask, bg_pre, border_active, only_tiles, exec_stages, am.a);         \
    } while (0)
    if (flags & MVX_FLAG_SPLIT_F16) { if (big) MVX_GO(2, 32, 2, 1); else MVX_GO(2, 32, 1, 1); }
    else if (np == 3) { if (big) MVX_GO(3, 16, 2, 0); else MVX_GO(3, 32, 1, 0); }
    else              { if (big) MVX_GO(2, 32, 2, 0); else MVX_GO(2, 32, 1, 0); }
#undef MVX_GO
}

// ------------------------------------------------------------------------------------------
// weight gradient, bf16x3.  dW[kd][a][b][c][n] = sum_sites x[site + tap][c] * dz[site][n]: the MFMA
// reduction index is the SITE, so both operands are needed "k-major" while memory is channel-major.
// The tiles are staged as [site][32 channels] bf16 rows (64 B) and fetched with ds_read_b64_tr_b16,
// the LDS transpose read: a 16-lane group reads a 4-site x 16-channel block and each lane receives
// 4 consecutive sites of its own channel -- exactly half of a 32x32x16 operand fragment, for any tap
// shift (the shift only changes which rows are addressed).  One 9-wave workgroup per (strip of
// patches, depth tap, 32-channel chunk); wave t owns in-plane tap t and a 32(c) x 64(n) accumulator.
// ------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
constexpr int WG_THREADS = 9 * 64;

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
__global__ __launch_bounds__(WG_THREADS) void conv3d_wgrad_split(const float *__restrict__ in,
                                                                 const float *__restrict__ dz,
                                                                 float *__restrict__ slabs, Geom g,
                                                                 int tiles_per_strip, const int *__restrict__ step_list,
                                                                 const int *__restrict__ step_count,
                                                                 const float *__restrict__ c_in, SplitAmax am) {
    float x_scale = 1.f, z_scale = 1.f;                       // fp16 pieces: operands scaled by their bound amax (split_common.h)
    if constexpr (FMT == 1) { x_scale = split_scale_coarse(am.a); z_scale = split_scale_of(am.b); }     // x: activations, dz: gradients
    __shared__ __attribute__((aligned(16))) unsigned short s_x[NP][HH * HW][BK];          // [piece][halo site][channel]
    __shared__ __attribute__((aligned(16))) unsigned short s_z[NP][2][TH * TW][32];       // [piece][32-channel half][site][channel]
    const int tiles_x = (g.W + TW - 1) / TW, tiles_y = (g.H + TH - 1) / TH;
    const int ntiles = tiles_x * tiles_y;
    const int strip = blockIdx.x;
    const int nchunks = g.Cin / BK;
    const int kd = blockIdx.y / nchunks, cc = blockIdx.y % nchunks;
    const int tid = threadIdx.x, lane = tid & 63, tap = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int ta = tap / 3, tb = tap % 3;
    // transpose-read roles of this lane
    const int grp = lane >> 4, i16 = lane & 15, q = i16 >> 2, pcol = (grp & 1) * 16 + 4 * (i16 & 3), kbase = (grp >> 1) * 8;

    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }

    // steps: dense = (valid plane) x (tile of the strip); background-aware = entries of the compacted list of this depth
    // tap dealt round-robin to the strips (see conv3d_wgrad4 in conv3d.hip)
    const int nstrips = gridDim.x;
    int dlo = 0, dhi = -1;
    for (int dd = 0; dd < g.Dout; ++dd) {
        const int ds = dd * g.sd - g.pd + kd;
        if (ds >= 0 && ds < g.Din) { if (dhi < 0) dlo = dd; dhi = dd; }
    }
    const int nd = dhi >= dlo ? dhi - dlo + 1 : 0;
    const int per = tiles_per_strip;
    const int *my_list = step_list ? step_list + (size_t)kd * g.Dout * g.F * ntiles : nullptr;
    const int nlist = step_list ? step_count[kd] : 0;
    const int nsteps = step_list ? (nlist > strip ? (nlist - strip + nstrips - 1) / nstrips : 0) : nd * per;
    auto step_of = [&](int i, int &d, int &t) {
        if (my_list) { const int e = my_list[strip + i * nstrips]; d = e / ntiles; t = e - d * ntiles; }
        else { d = dlo + i / per; t = strip * per + i % per; }
    };
    auto next_live = [&](int i) {
        if (!my_list)
            while (i < nsteps && strip * per + i % per >= ntiles) ++i;
        return i < nsteps ? i : nsteps;
    };
    constexpr int NX = (HH * HW * 8 + WG_THREADS - 1) / WG_THREADS;
    constexpr int NZ = (TH * TW * 16 + WG_THREADS - 1) / WG_THREADS;
    float4 xr[NX], zr[NZ];
    auto load_step = [&](int i) {
        int d, t;
        step_of(i, d, t);
        const int ds = src_depth(g, d, kd);                       // listed / dense steps always have a valid source plane
        const int tx0 = (t % tiles_x) * TW, ty0 = (t / tiles_x) * TH;
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            const int c = tid + WG_THREADS * u;
            xr[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < HH * HW * 8) {
                const int r = c >> 3, part = c & 7;
                const int gy = ty0 - 1 + r / HW, gx = tx0 - 1 + r % HW;
                if (gy >= 0 && gy < g.H && gx >= 0 && gx < g.W) {
                    float4 v = *(const float4 *)(in + (((size_t)ds * g.H + gy) * g.W + gx) * g.Cin + cc * BK + part * 4);
                    if (c_in) {
                        const float4 cb = *(const float4 *)(c_in + (size_t)ds * g.Cin + cc * BK + part * 4);
                        v.x -= cb.x; v.y -= cb.y; v.z -= cb.z; v.w -= cb.w;
                    }
                    xr[u] = v;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < NZ; ++u) {
            const int c = tid + WG_THREADS * u;
            zr[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < TH * TW * 16) {
                const int r = c >> 4, part = c & 15;
                const int gy = ty0 + (r >> 4), gx = tx0 + (r & 15);
                if (gy < g.H && gx < g.W)
                    zr[u] = *(const float4 *)(dz + (((size_t)d * g.H + gy) * g.W + gx) * g.Cout + part * 4);
            }
        }
    };
    int cur = next_live(0);
    if (cur < nsteps) load_step(cur);
    while (cur < nsteps) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            const int c = tid + WG_THREADS * u;
            if (c < HH * HW * 8) {
                uint2 pc[NP];
                if constexpr (FMT == 1) { xr[u].x *= x_scale; xr[u].y *= x_scale; xr[u].z *= x_scale; xr[u].w *= x_scale; }
                split_n<NP, FMT>(xr[u].x, xr[u].y, xr[u].z, xr[u].w, pc);
#pragma unroll
                for (int p = 0; p < NP; ++p) *(uint2 *)(&s_x[p][c >> 3][(c & 7) * 4]) = pc[p];
            }
        }
#pragma unroll
        for (int u = 0; u < NZ; ++u) {
            const int c = tid + WG_THREADS * u;
            if (c < TH * TW * 16) {
                const int r = c >> 4, part = c & 15;
                uint2 pc[NP];
                if constexpr (FMT == 1) { zr[u].x *= z_scale; zr[u].y *= z_scale; zr[u].z *= z_scale; zr[u].w *= z_scale; }
                split_n<NP, FMT>(zr[u].x, zr[u].y, zr[u].z, zr[u].w, pc);
#pragma unroll
                for (int p = 0; p < NP; ++p) *(uint2 *)(&s_z[p][part >> 3][r][(part & 7) * 4]) = pc[p];
            }
        }
        __syncthreads();
        const int nxt = next_live(cur + 1);
        if (nxt < nsteps) load_step(nxt);
#pragma unroll 2
        for (int ks = 0; ks < TH; ++ks) {                 // 16 sites (one patch row) per MFMA k-step
            const int hr0 = (ks + ta) * HW + tb + kbase + q, hr1 = hr0 + 4;      // halo rows of sites kbase+q, +4
            const int zr0 = ks * TW + kbase + q, zr1 = zr0 + 4;
            bf16x8 av[NP], b0[NP], b1[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                av[p] = tr_frag(&s_x[p][hr0][pcol], &s_x[p][hr1][pcol]);
                b0[p] = tr_frag(&s_z[p][0][zr0][pcol], &s_z[p][0][zr1][pcol]);
                b1[p] = tr_frag(&s_z[p][1][zr0][pcol], &s_z[p][1][zr1][pcol]);
            }
            split_mac2<NP, FMT>(acc0, acc1, av, b0, b1);
        }
        cur = nxt;
    }
    float *o = slabs + ((((size_t)strip * 3 + kd) * 9 + tap) * g.Cin + cc * BK) * BN;
    if constexpr (FMT == 1) {
        const float o_scale = split_inverse(x_scale) * split_inverse(z_scale);
        acc0 *= o_scale; acc1 *= o_scale;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        o[(size_t)row * BN + li] = acc0[r];
        o[(size_t)row * BN + 32 + li] = acc1[r];
    }
}

__global__ void wgrad_reduce_split(const float *__restrict__ slabs, float *__restrict__ dw, int nstrips, int Ci, int accumulate) {
    const size_t per = (size_t)27 * Ci * BN;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < per; e += (size_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < nstrips; ++k) s += slabs[(size_t)k * per + e];
        const int co = (int)(e % BN);
        size_t r = e / BN;
        const int ci = (int)(r % Ci); r /= Ci;
        const int tap = (int)(r % 9);
        const int kd = (int)(r / 9);
        float *dst = dw + ((((size_t)co * Ci + ci) * 3 + kd) * 3 + tap / 3) * 3 + tap % 3;
        *dst = accumulate ? *dst + s : s;
    }
}

int check_geom(int32_t din, int32_t dout, int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t sd, int32_t pd) {
    // a rejected call drops the operand ranges bound for it (common.h MVX_CHECK_ARG)
    MVX_CHECK_ARG(!(din <= 0 || dout <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0));
    MVX_CHECK_ARG(!(sd < 1 || sd > 2 || pd < 0 || pd > 1));
    if (cin % BK || cout % BN) {
        mvxi_drop_split_amax();
        return MVX_ESIZE;
    }
    return MVX_OK;
}

static inline int pieces_of(int flags) { return (flags & MVX_FLAG_SPLIT_F16) ? 2 : (flags & MVX_FLAG_SPLIT3) ? 3 : 2; }
static inline int fmt_of(int flags) { return (flags & MVX_FLAG_SPLIT_F16) ? 1 : 0; }

}  // namespace

extern "C" size_t mvx_conv3d_packed_weight_bytes_split(int32_t cout, int32_t cin, int32_t flags) {
    if (cout <= 0 || cin <= 0) return 0;
    return (size_t)27 * cout * cin * pieces_of(flags) * sizeof(unsigned short);
}

extern "C" int mvx_conv3d_pack_weights_split(const float *w, void *wsplit, int32_t cout, int32_t cin, int32_t for_dgrad,
                                             int32_t flags, void *stream) {
    MVX_CHECK_ARG(w && wsplit && cout > 0 && cin > 0);
    MVX_CHECK_ARG((for_dgrad ? cout : cin) % BK == 0);
    const long long total = 27ll * cout * cin;
    hipLaunchKernelGGL(pack_weights_split, dim3(mvx_cdiv(total, 256) > 2048 ? 2048 : mvx_cdiv(total, 256)), dim3(256), 0,
                       (hipStream_t)stream, w, (unsigned short *)wsplit, cout, cin, for_dgrad, pieces_of(flags), fmt_of(flags));
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" int mvx_conv3d_forward_split(const float *in, const void *wsplit, const float *bias, float *out, double *stats,
                                        int32_t din, int32_t dout, int32_t h, int32_t w, int32_t cin, int32_t cout,
                                        int32_t stride_d, int32_t pad_d, int32_t flags, void *stream) {
    MVX_CHECK_ARG(in && wsplit && out);
    const int relu = flags & MVX_FLAG_RELU;
    int rc = check_geom(din, dout, h, w, cin, cout, stride_d, pad_d);
    if (rc) return rc;
    MVX_CHECK_ARG(dout == (din + 2 * pad_d - 3) / stride_d + 1);
    hipStream_t st = (hipStream_t)stream;
    if (stats && !(flags & MVX_FLAG_PREZEROED)) {
        hipError_t e = hipMemsetAsync(stats, 0, sizeof(double) * MVX_REP * 2 * cout, st);
        if (e != hipSuccess) return (int)e;
    }
    Geom g{din, dout, h, w, cin, cout, stride_d, pad_d, 0};
    launch_gather_split(st, flags, dout, cout / BN, in, (const unsigned short *)wsplit, bias, out, stats, g, relu, nullptr, nullptr, nullptr,
                        0, nullptr, nullptr);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" int mvx_conv3d_forward_bg_split_frames(const float *in, const void *wsplit, const float *bias, float *out,
                                                  double *stats, int32_t din, int32_t dout, int32_t h, int32_t w, int32_t cin,
                                                  int32_t cout, int32_t stride_d, int32_t pad_d, int32_t flags,
                                                  const int32_t *in_halo_flags, const uint8_t *out_mask, const float *bg_pre,
                                                  int32_t border_active, uint64_t *exec_stages, int32_t n_frames, void *stream) {
    MVX_CHECK_ARG(in && wsplit && out && in_halo_flags && out_mask && bg_pre);
    MVX_CHECK_ARG(n_frames >= 1 && n_frames <= MVX_MAX_FRAMES);
    int rc = check_geom(din, dout, h, w, cin, cout, stride_d, pad_d);
    if (rc) return rc;
    MVX_CHECK_ARG(dout == (din + 2 * pad_d - 3) / stride_d + 1);
    hipStream_t st = (hipStream_t)stream;
    if (stats && !(flags & MVX_FLAG_PREZEROED)) {
        hipError_t e = hipMemsetAsync(stats, 0, sizeof(double) * MVX_REP * 2 * cout * n_frames, st);
        if (e != hipSuccess) return (int)e;
    }
    Geom g{din, dout, h, w, cin, cout, stride_d, pad_d, 0, n_frames};
    launch_gather_split(st, flags, dout * n_frames, cout / BN, in, (const unsigned short *)wsplit, bias, out, stats, g,
                        flags & MVX_FLAG_RELU, in_halo_flags, out_mask, bg_pre,
                        (border_active ? 1 : 0) | ((flags & MVX_FLAG_BG_TAPS) ? 2 : 0), nullptr, (unsigned long long *)exec_stages);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" int mvx_conv3d_forward_bg_split(const float *in, const void *wsplit, const float *bias, float *out, double *stats,
                                           int32_t din, int32_t dout, int32_t h, int32_t w, int32_t cin, int32_t cout,
                                           int32_t stride_d, int32_t pad_d, int32_t flags, const int32_t *in_halo_flags,
                                           const uint8_t *out_mask, const float *bg_pre, int32_t border_active,
                                           void *stream) {
    return mvx_conv3d_forward_bg_split_frames(in, wsplit, bias, out, stats, din, dout, h, w, cin, cout, stride_d, pad_d, flags,
                                              in_halo_flags, out_mask, bg_pre, border_active, nullptr, 1, stream);
}

static int launch_dgrad_split(const float *dz, const void *wsplit_dgrad, float *dx, int32_t din, int32_t dout, int32_t h,
                              int32_t w, int32_t cin, int32_t cout, int32_t stride_d, int32_t pad_d, int32_t flags,
                              const int32_t *only_tiles, uint64_t *exec_stages, int32_t n_frames, void *stream) {
    MVX_CHECK_ARG(dz && wsplit_dgrad && dx);
    MVX_CHECK_ARG(n_frames >= 1 && n_frames <= MVX_MAX_FRAMES);
    int rc = check_geom(din, dout, h, w, cout, cin, stride_d, pad_d);
    if (rc) return rc;
    Geom g{dout, din, h, w, cout, cin, stride_d, pad_d, 1, n_frames};
    if (flags & MVX_FLAG_TAPS2) {                   // the input gradient of such a layer reads the flipped window {1,2}^2
        g.tap_lo = 1; g.tap_hi = 3;
        if (cin % 4 == 0 && (cin / 4) % BN == 0) g.s2d = cin / 4;      // parity of the OUTPUT channel block
    }
    launch_gather_split((hipStream_t)stream, flags, din * n_frames, cin / BN, dz, (const unsigned short *)wsplit_dgrad, nullptr, dx, nullptr,
                        g, 0, nullptr, nullptr, nullptr, 0, only_tiles, (unsigned long long *)exec_stages);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" int mvx_conv3d_dgrad_tiles_split_frames(const float *dz, const void *wsplit_dgrad, float *dx, int32_t din,
                                                   int32_t dout, int32_t h, int32_t w, int32_t cin, int32_t cout,
                                                   int32_t stride_d, int32_t pad_d, int32_t flags,
                                                   const int32_t *dx_tile_flags, uint64_t *exec_stages, int32_t n_frames,
                                                   void *stream) {
    MVX_CHECK_ARG(dx_tile_flags);
    return launch_dgrad_split(dz, wsplit_dgrad, dx, din, dout, h, w, cin, cout, stride_d, pad_d, flags, dx_tile_flags,
                              exec_stages, n_frames, stream);
}

extern "C" int mvx_conv3d_dgrad_tiles_split(const float *dz, const void *wsplit_dgrad, float *dx, int32_t din, int32_t dout,
                                            int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t stride_d,
                                            int32_t pad_d, int32_t flags, const int32_t *dx_tile_flags, void *stream) {
    MVX_CHECK_ARG(dx_tile_flags);
    return launch_dgrad_split(dz, wsplit_dgrad, dx, din, dout, h, w, cin, cout, stride_d, pad_d, flags, dx_tile_flags, nullptr, 1,
                              stream);
}

extern "C" int mvx_conv3d_dgrad_split(const float *dz, const void *wsplit_dgrad, float *dx, int32_t din, int32_t dout,
                                      int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t stride_d, int32_t pad_d,
                                      int32_t flags, void *stream) {
    return launch_dgrad_split(dz, wsplit_dgrad, dx, din, dout, h, w, cin, cout, stride_d, pad_d, flags, nullptr, nullptr, 1, stream);
}

extern "C" int mvx_conv3d_wgrad_split(const float *in, const float *dz, float *dw, int32_t din, int32_t dout, int32_t h,
                                      int32_t w, int32_t cin, int32_t cout, int32_t stride_d, int32_t pad_d,
                                      int32_t flags, void *workspace, size_t workspace_bytes, void *stream) {
    const SplitAmax am = mvxi_take_split_amax();         // (x, dz) of an fp16-piece launch, else NULLs
    MVX_CHECK_ARG(in && dz && dw && workspace);
    int rc = check_geom(din, dout, h, w, cin, cout, stride_d, pad_d);
    if (rc) return rc;
    if (cout != BN) return MVX_ESIZE;
    // same strip decomposition (and workspace size) as mvx_conv3d_wgrad
    const int ntiles = (int)(mvx_cdiv(w, TW) * mvx_cdiv(h, TH));
    int strips = 512 / (3 * (cin / BK));             // one full round of 2 workgroups per CU (see conv3d.hip)
    if (strips < 1) strips = 1;
    int per = (ntiles + strips - 1) / strips;
    if (per < 1) per = 1;
    const int nstrips = (ntiles + per - 1) / per;
    MVX_CHECK_ARG(workspace_bytes >= (size_t)nstrips * 27 * cin * BN * sizeof(float));
    Geom g{din, dout, h, w, cin, cout, stride_d, pad_d, 0};
    hipStream_t st = (hipStream_t)stream;
    if (fmt_of(flags))
        hipLaunchKernelGGL((conv3d_wgrad_split<2, 1>), dim3(nstrips, 3 * (cin / BK)), dim3(WG_THREADS), 0, st, in, dz,
                           (float *)workspace, g, per, (const int *)nullptr, (const int *)nullptr, (const float *)nullptr, am);
    else if (pieces_of(flags) == 3)
        hipLaunchKernelGGL((conv3d_wgrad_split<3, 0>), dim3(nstrips, 3 * (cin / BK)), dim3(WG_THREADS), 0, st, in, dz,
                           (float *)workspace, g, per, (const int *)nullptr, (const int *)nullptr, (const float *)nullptr, am);
    else
        hipLaunchKernelGGL((conv3d_wgrad_split<2, 0>), dim3(nstrips, 3 * (cin / BK)), dim3(WG_THREADS), 0, st, in, dz,
                           (float *)workspace, g, per, (const int *)nullptr, (const int *)nullptr, (const float *)nullptr, am);
    MVX_LAUNCH_CHECK();
    const size_t per_slab = (size_t)27 * cin * BN;
    hipLaunchKernelGGL(wgrad_reduce_split, dim3(mvx_cdiv(per_slab, 256)), dim3(256), 0, st, (const float *)workspace, dw,
                       nstrips, cin, flags & MVX_FLAG_ACCUMULATE);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

// background-aware weight gradient, bf16x3: same decomposition as mvx_conv3d_wgrad_bg (conv3d.hip)
static int wgrad_bg_split_strips(int cin) {
    const int s = 256 / (3 * (cin / BK));
    return s < 1 ? 1 : s;
}

extern "C" size_t mvx_conv3d_wgrad_bg_split_workspace_bytes_frames(int32_t dout, int32_t h, int32_t w, int32_t cin, int32_t cout,
                                                                  int32_t n_frames) {
    if (dout <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout != BN || cin % BK || n_frames <= 0) return 0;
    const size_t ntiles = (size_t)mvx_cdiv(w, TW) * mvx_cdiv(h, TH);
    return (size_t)wgrad_bg_split_strips(cin) * 27 * cin * BN * sizeof(float) + sizeof(int) * (3 * dout * n_frames * ntiles + 4);
}

extern "C" size_t mvx_conv3d_wgrad_bg_split_workspace_bytes(int32_t dout, int32_t h, int32_t w, int32_t cin, int32_t cout) {
    return mvx_conv3d_wgrad_bg_split_workspace_bytes_frames(dout, h, w, cin, cout, 1);
}

extern "C" int mvx_conv3d_wgrad_bg_split_frames(const float *in, const float *dz, float *dw, int32_t din, int32_t dout,
                                                int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t stride_d,
                                                int32_t pad_d, int32_t flags, const int32_t *in_halo_flags, const float *c_in,
                                                const float *tap_sums, void *workspace, size_t workspace_bytes,
                                                int32_t n_frames, void *stream) {
    const SplitAmax am = mvxi_take_split_amax();         // (x, dz) of an fp16-piece launch, else NULLs
    MVX_CHECK_ARG(in && dz && dw && workspace && in_halo_flags && c_in && tap_sums);
    MVX_CHECK_ARG(n_frames >= 1 && n_frames <= MVX_MAX_FRAMES);
    int rc = check_geom(din, dout, h, w, cin, cout, stride_d, pad_d);
    if (rc) return rc;
    if (cout != BN) return MVX_ESIZE;
    MVX_CHECK_ARG(workspace_bytes >= mvx_conv3d_wgrad_bg_split_workspace_bytes_frames(dout, h, w, cin, cout, n_frames));
    const int ntiles = (int)(mvx_cdiv(w, TW) * mvx_cdiv(h, TH));
    const int nstrips = wgrad_bg_split_strips(cin);
    Geom g{din, dout, h, w, cin, cout, stride_d, pad_d, 0, n_frames};
    hipStream_t st = (hipStream_t)stream;
    float *slabs = (float *)workspace;
    int *list = (int *)((char *)workspace + (size_t)nstrips * 27 * cin * BN * sizeof(float));
    int *count = list + (size_t)3 * dout * n_frames * ntiles;
    rc = mvxi_wgrad_step_list(in_halo_flags, din, dout, ntiles, stride_d, pad_d, list, count, st, n_frames);
    if (rc) return rc;
    if (fmt_of(flags))
        hipLaunchKernelGGL((conv3d_wgrad_split<2, 1>), dim3(nstrips, 3 * (cin / BK)), dim3(WG_THREADS), 0, st, in, dz, slabs, g, 0,
                           (const int *)list, (const int *)count, c_in, am);
    else if (pieces_of(flags) == 3)
        hipLaunchKernelGGL((conv3d_wgrad_split<3, 0>), dim3(nstrips, 3 * (cin / BK)), dim3(WG_THREADS), 0, st, in, dz, slabs, g, 0,
                           (const int *)list, (const int *)count, c_in, am);
    else
        hipLaunchKernelGGL((conv3d_wgrad_split<2, 0>), dim3(nstrips, 3 * (cin / BK)), dim3(WG_THREADS), 0, st, in, dz, slabs, g, 0,
                           (const int *)list, (const int *)count, c_in, am);
    MVX_LAUNCH_CHECK();
    const size_t per_slab = (size_t)27 * cin * BN;
    hipLaunchKernelGGL(wgrad_reduce_split, dim3(mvx_cdiv(per_slab, 256)), dim3(256), 0, st, (const float *)slabs, dw, nstrips,
                       cin, flags & MVX_FLAG_ACCUMULATE);
    MVX_LAUNCH_CHECK();
    return mvxi_wgrad_rank1(tap_sums, c_in, dw, din, dout, cin, cout, stride_d, pad_d, st, n_frames);
}

extern "C" int mvx_conv3d_wgrad_bg_split(const float *in, const float *dz, float *dw, int32_t din, int32_t dout, int32_t h,
                                         int32_t w, int32_t cin, int32_t cout, int32_t stride_d, int32_t pad_d,
                                         int32_t flags, const int32_t *in_halo_flags, const float *c_in,
                                         const float *tap_sums, void *workspace, size_t workspace_bytes, void *stream) {
    return mvx_conv3d_wgrad_bg_split_frames(in, dz, dw, din, dout, h, w, cin, cout, stride_d, pad_d, flags, in_halo_flags, c_in,
                                            tap_sums, workspace, workspace_bytes, 1, stream);
}

// ------------------------------------------------------------------------------------------
// 2-D convolutions of the RPN on frame sets, bf16x3 (modules/voxelnet/Pipe.py:45-75): the same kernels with one plane
// per frame (din = dout = 1, pad_d = 1: only the middle depth tap exists).  Weights: the 2-D kernel placed in the middle
// depth slice of a 3-D one and packed with mvx_conv3d_pack_weights_split.  The stride-2 layers run on the space-to-depth
// image with their rearranged 3x3 kernel (zeros where the 2x2 window has no tap: all nine taps are executed here).
// ------------------------------------------------------------------------------------------
extern "C" int mvx_conv2d_forward_split_frames(const float *in, const void *wsplit, const float *bias, float *out, double *stats,
                                               int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t flags,
                                               int32_t n_frames, void *stream) {
    MVX_CHECK_ARG(in && wsplit && out);
    MVX_CHECK_ARG(n_frames >= 1 && n_frames <= MVX_MAX_FRAMES);
    int rc = check_geom(1, 1, h, w, cin, cout, 1, 1);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (stats && !(flags & MVX_FLAG_PREZEROED)) {
        hipError_t e = hipMemsetAsync(stats, 0, sizeof(double) * MVX_REP * 2 * cout * n_frames, st);
        if (e != hipSuccess) return (int)e;
    }
    Geom g{1, 1, h, w, cin, cout, 1, 1, 0, n_frames};
    if (flags & MVX_FLAG_TAPS2) {                   // stride-2 kernel on the space-to-depth image: window {0,1}^2, structural zeros skipped
        g.tap_lo = 0; g.tap_hi = 2;
        if (cin % 4 == 0 && (cin / 4) % BK == 0) g.s2d = cin / 4;
    }
    launch_gather_split(st, flags, n_frames, cout / BN, in, (const unsigned short *)wsplit, bias, out, stats, g, flags & MVX_FLAG_RELU,
                        nullptr, nullptr, nullptr, 0, nullptr, nullptr);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" int mvx_conv2d_dgrad_split_frames(const float *dz, const void *wsplit_dgrad, float *dx, int32_t h, int32_t w,
                                             int32_t cin, int32_t cout, int32_t flags, int32_t n_frames, void *stream) {
    return launch_dgrad_split(dz, wsplit_dgrad, dx, 1, 1, h, w, cin, cout, 1, 1, flags, nullptr, nullptr, n_frames, stream);
}

static int conv2d_wgrad_split_strips(int cin) {
    const int s = 256 / (cin / BK);                 // one depth tap carries work: 256 workgroups per 64-channel block of dz
    return s < 1 ? 1 : (s > 64 ? 64 : s);
}


extern "C" size_t mvx_conv2d_wgrad_split_workspace_bytes_frames(int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t n_frames) {
    if (h <= 0 || w <= 0 || cin <= 0 || cout <= 0 || cin % BK || cout % BN || n_frames <= 0) return 0;
    const size_t ntiles = (size_t)mvx_cdiv(w, TW) * mvx_cdiv(h, TH);
    return (size_t)conv2d_wgrad_split_strips(cin) * 27 * cin * BN * sizeof(float) +
           sizeof(int) * (3 * (size_t)n_frames * ntiles + 4) + sizeof(int) * (size_t)n_frames * ntiles;
}

// dw3 f32 [cout][cin][3][3][3]: the 2-D gradient is its middle depth slice (slices 0 and 2 come out zero)
extern "C" int mvx_conv2d_wgrad_split_frames(const float *in, const float *dz, float *dw3, int32_t h, int32_t w, int32_t cin,
                                             int32_t cout, int32_t flags, void *workspace, size_t workspace_bytes,
                                             int32_t n_frames, void *stream) {
    const SplitAmax am = mvxi_take_split_amax();         // (x, dz) of an fp16-piece launch, else NULLs
    MVX_CHECK_ARG(in && dz && dw3 && workspace);
    MVX_CHECK_ARG(n_frames >= 1 && n_frames <= MVX_MAX_FRAMES);
    int rc = check_geom(1, 1, h, w, cin, cout, 1, 1);
    if (rc) return rc;
    MVX_CHECK_ARG(workspace_bytes >= mvx_conv2d_wgrad_split_workspace_bytes_frames(h, w, cin, cout, n_frames));
    hipStream_t st = (hipStream_t)stream;
    const int ntiles = (int)(mvx_cdiv(w, TW) * mvx_cdiv(h, TH));
    const int nstrips = conv2d_wgrad_split_strips(cin);
    float *slabs = (float *)workspace;
    int *list = (int *)((char *)workspace + (size_t)nstrips * 27 * cin * BN * sizeof(float));
    int *count = list + (size_t)3 * n_frames * ntiles;
    int *ones = count + 4;                          // "every tile is a step": any non-zero word is a set flag
    hipError_t e = hipMemsetAsync(ones, 0x01, sizeof(int) * (size_t)n_frames * ntiles, st);
    if (e != hipSuccess) return (int)e;
    rc = mvxi_wgrad_step_list(ones, 1, 1, ntiles, 1, 1, list, count, st, n_frames);
    if (rc) return rc;
    Geom g{1, 1, h, w, cin, cout, 1, 1, 0, n_frames};
    const size_t per_slab = (size_t)27 * cin * BN;
    for (int nb = 0; nb < cout / BN; ++nb) {        // the kernel owns 64 channels of dz per launch
        if (fmt_of(flags))
            hipLaunchKernelGGL((conv3d_wgrad_split<2, 1>), dim3(nstrips, 3 * (cin / BK)), dim3(WG_THREADS), 0, st, in,
                               dz + (size_t)nb * BN, slabs, g, 0, (const int *)list, (const int *)count, (const float *)nullptr, am);
        else if (pieces_of(flags) == 3)
            hipLaunchKernelGGL((conv3d_wgrad_split<3, 0>), dim3(nstrips, 3 * (cin / BK)), dim3(WG_THREADS), 0, st, in,
                               dz + (size_t)nb * BN, slabs, g, 0, (const int *)list, (const int *)count, (const float *)nullptr, am);
        else
            hipLaunchKernelGGL((conv3d_wgrad_split<2, 0>), dim3(nstrips, 3 * (cin / BK)), dim3(WG_THREADS), 0, st, in,
                               dz + (size_t)nb * BN, slabs, g, 0, (const int *)list, (const int *)count, (const float *)nullptr, am);
        MVX_LAUNCH_CHECK();
        hipLaunchKernelGGL(wgrad_reduce_split, dim3(mvx_cdiv(per_slab, 256)), dim3(256), 0, st, (const float *)slabs,
                           dw3 + (size_t)nb * BN * cin * 27, nstrips, cin, 0);
        MVX_LAUNCH_CHECK();
    }
    return MVX_OK;
}
