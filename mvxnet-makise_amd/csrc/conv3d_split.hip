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
//   - LDS rows of 144 bytes: 32 hi (64 B) | 32 lo (64 B) | 16 B pad -> ds_read_b128 operand fetches
//     (8 consecutive k per lane, the native 32x32x16 A/B fragment) are bank-conflict free;
//   - the f32 -> (hi, lo) split of the activations happens while the halo is staged;
//   - weights are pre-split by the pack kernel and staged three taps (one kernel row) at a time, so a
//     barrier pair covers 3 taps x 12 MFMAs per wave instead of one tap.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int TH = 8, TW = 16, HH = TH + 2, HW = TW + 2;
constexpr int BK = 32, BN = 64;
constexpr int ROWB = 144;                         // bytes per LDS row: 64 hi + 64 lo + 16 pad

struct Geom {
    int Din, Dout, H, W, Cin, Cout, sd, pd, mode;
    int F = 1;                                     // frames stacked along depth: planes [F * Dout] <- [F * Din]
};

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

__device__ __forceinline__ void split4(const float4 v, uint2 *hi, uint2 *lo) {
    // hi = bf16(x) (round to nearest even), lo = bf16(x - hi)
    const float x[4] = {v.x, v.y, v.z, v.w};
    unsigned short h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const __bf16 hb = (__bf16)x[j];
        const __bf16 lb = (__bf16)(x[j] - (float)hb);
        h[j] = __builtin_bit_cast(unsigned short, hb);
        l[j] = __builtin_bit_cast(unsigned short, lb);
    }
    hi->x = (unsigned)h[0] | ((unsigned)h[1] << 16); hi->y = (unsigned)h[2] | ((unsigned)h[3] << 16);
    lo->x = (unsigned)l[0] | ((unsigned)l[1] << 16); lo->y = (unsigned)l[2] | ((unsigned)l[3] << 16);
}

// torch W[co][ci][kd][kh][kw] -> wsp[kd][tap][chunk][n][hi 32 | lo 32] (bf16), forward or dgrad view
__global__ void pack_weights_split(const float *__restrict__ w, unsigned short *__restrict__ wsp, int Co, int Ci, int dgrad) {
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
        const float x = w[((((long long)co * Ci + ci) * 3 + kd) * 3 + kh) * 3 + kw];
        const __bf16 hb = (__bf16)x;
        const __bf16 lb = (__bf16)(x - (float)hb);
        const long long row = (((long long)kd * 9 + tap) * nch + ch) * N + n;
        wsp[row * 2 * BK + k] = __builtin_bit_cast(unsigned short, hb);
        wsp[row * 2 * BK + BK + k] = __builtin_bit_cast(unsigned short, lb);
    }
}

__global__ __launch_bounds__(256, 2) void conv3d_gather_split(const float *__restrict__ in,
                                                              const unsigned short *__restrict__ wsp,
                                                              const float *__restrict__ bias,
                                                              float *__restrict__ out, double *__restrict__ stats,
                                                              Geom g, int relu, const int *__restrict__ in_hflag,
                                                              const unsigned char *__restrict__ out_mask,
                                                              const float *__restrict__ bg_pre, int border_active,
                                                              const int *__restrict__ only_tiles,
                                                              unsigned long long *__restrict__ exec_stages) {
    __shared__ __attribute__((aligned(16))) unsigned char s_halo[HH * HW * ROWB];
    __shared__ __attribute__((aligned(16))) unsigned char s_w[3][BN * ROWB];
    __shared__ float s_red[4][2 * BN];
    const int tiles_x = (g.W + TW - 1) / TW;
    const int tx0 = (blockIdx.x % tiles_x) * TW, ty0 = (blockIdx.x / tiles_x) * TH;
    const int d = blockIdx.y, nb = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int nchunks = g.Cin / BK;
    // background rewrite (see conv3d.hip / activity.hip): restricted launches and constant tiles
    if (only_tiles && !only_tiles[(size_t)d * gridDim.x + blockIdx.x]) return;
    // border_active bit 0: border tiles always count as active; bit 1: bg_pre carries the per-depth-tap and position-class
    // constants (see gather_unit in conv3d.hip): interior tiles skip depth taps with a background-only source halo, border
    // tiles without any active source are filled from the class constants
    const bool on_border = tx0 == 0 || ty0 == 0 || tx0 + TW >= g.W || ty0 + TH >= g.H;
    const bool skip_taps = in_hflag && (border_active & 2) && !on_border;
    unsigned skipped = 0;
    int any_flag = 0;
    bool active = true;
    if (in_hflag) {
        for (int kd = 0; kd < 3; ++kd) {
            const int ds = src_depth(g, d, kd);
            if (ds >= 0) {
                const int fl = in_hflag[(size_t)ds * gridDim.x + blockIdx.x];
                any_flag |= fl;
                if (skip_taps && !fl) skipped |= 1u << kd;
            }
        }
        active = (((border_active & 1) && on_border) || any_flag) != 0;
    }
    const bool idle_border = in_hflag && (border_active & 2) && on_border && !any_flag;

    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const int my_ty = 2 * wv + (li >> 4), my_tx = li & 15;
    const int a_base = (my_ty * HW + my_tx) * ROWB + lh * 16;     // bytes; + part*64 + kstep*32
    const int b_base = li * ROWB + lh * 16;

    // three weight tiles (one kernel row) = 3 x 8 KB = 1536 x 16 B -> 6 per thread.  A NATIVE vector type: an array of HIP's
    // uint4 struct stays an alloca, and the backend "promoted" it to 24 KB of LDS (80 KB per workgroup: one workgroup per CU)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 wreg[6];
    auto load_w3 = [&](int kd, int a, int cc) {
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int c = tid + 256 * u;              // 0..1535
            const int t = c >> 9, rem = c & 511;      // tap in row, 16-byte piece of the tile
            const unsigned char *tile = (const unsigned char *)wsp +
                ((((size_t)kd * 9 + a * 3 + t) * nchunks + cc) * g.Cout + (size_t)nb * BN) * (2 * BK * 2);
            wreg[u] = *(const u32x4 *)(tile + (size_t)rem * 16);
        }
    };
    auto store_w3 = [&]() {
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int c = tid + 256 * u;
            const int t = c >> 9, rem = c & 511;
            const int n = rem >> 3, piece = rem & 7;  // 8 pieces of 16 B per 128-B row
            *(u32x4 *)(s_w[t] + n * ROWB + piece * 16) = wreg[u];
        }
    };

    // Stages = (valid depth tap) x (32-channel chunk), software-pipelined like conv3d_gather_pf (conv3d.hip): the halo of
    // stage s + 1 is fetched into registers while stage s computes (split into hi / lo when it is written to LDS), the
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
    if (exec_stages && nstages > 0 && threadIdx.x == 0) atomicAdd(exec_stages, (unsigned long long)nstages);    // executed stages only
    auto stage_of = [&](int st, int &kd, int &ds, int &cc) __attribute__((always_inline)) {
        const int i = st / nchunks;
        cc = st - i * nchunks;
        kd = i == 0 ? kd_l[0] : (i == 1 ? kd_l[1] : kd_l[2]);
        ds = i == 0 ? ds_l[0] : (i == 1 ? ds_l[1] : ds_l[2]);
    };
    int h_off[6], h_lds[6];     // per-thread halo slots: global float offset inside a (plane, chunk) image or -1; LDS byte offset or -1
#pragma unroll
    for (int u = 0; u < 6; ++u) {
        const int c = tid + 256 * u;
        h_off[u] = -1;
        h_lds[u] = -1;
        if (c < HH * HW * 8) {
            const int r = c >> 3, part = c & 7;
            const int gy = ty0 - 1 + r / HW, gx = tx0 - 1 + r % HW;
            h_lds[u] = r * ROWB + part * 8;
            if (gy >= 0 && gy < g.H && gx >= 0 && gx < g.W) h_off[u] = (gy * g.W + gx) * g.Cin + part * 4;
        }
    }
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 hreg[6];
    auto load_halo = [&](int st) __attribute__((always_inline)) {
        int kd, ds, cc;
        stage_of(st, kd, ds, cc);
        const float *img = in + (size_t)ds * g.H * g.W * g.Cin + cc * BK;
#pragma unroll
        for (int u = 0; u < 6; ++u)
            hreg[u] = h_off[u] >= 0 ? *(const f32x4 *)(img + h_off[u]) : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto store_halo = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 6; ++u)
            if (h_lds[u] >= 0) {
                uint2 hi, lo;
                split4(make_float4(hreg[u][0], hreg[u][1], hreg[u][2], hreg[u][3]), &hi, &lo);
                *(uint2 *)(s_halo + h_lds[u]) = hi;
                *(uint2 *)(s_halo + h_lds[u] + 64) = lo;
            }
    };
    auto load_wrow = [&](int st, int a) __attribute__((always_inline)) {
        int kd, ds, cc;
        stage_of(st, kd, ds, cc);
        load_w3(kd, a, cc);
    };
    auto compute_row = [&](int a) __attribute__((always_inline)) {
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const int a_off = a_base + (a * HW + b) * ROWB;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 ah = __builtin_bit_cast(bf16x8, *(const uint4 *)(s_halo + a_off + s2 * 32));
                const bf16x8 al = __builtin_bit_cast(bf16x8, *(const uint4 *)(s_halo + a_off + 64 + s2 * 32));
                const bf16x8 b0h = __builtin_bit_cast(bf16x8, *(const uint4 *)(s_w[b] + b_base + s2 * 32));
                const bf16x8 b0l = __builtin_bit_cast(bf16x8, *(const uint4 *)(s_w[b] + b_base + 64 + s2 * 32));
                const bf16x8 b1h = __builtin_bit_cast(bf16x8, *(const uint4 *)(s_w[b] + b_base + 32 * ROWB + s2 * 32));
                const bf16x8 b1l = __builtin_bit_cast(bf16x8, *(const uint4 *)(s_w[b] + b_base + 32 * ROWB + 64 + s2 * 32));
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b0h, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b1h, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b0l, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b1l, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b0h, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b1h, acc1, 0, 0, 0);
            }
        }
    };
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
        compute_row(0);
        __syncthreads();
        store_w3();                                           // kernel row 1
        __syncthreads();
        load_wrow(st, 2);
        compute_row(1);
        __syncthreads();
        store_w3();                                           // kernel row 2
        __syncthreads();
        load_wrow(nxt, 0);
        compute_row(2);
    }

    const int n0 = nb * BN + li, n1 = n0 + 32;
    const float bias0 = bias ? bias[n0] : 0.f, bias1 = bias ? bias[n1] : 0.f;
    if (idle_border) {                         // the accumulators are still zero: they take the position-class constants
        const float *bg_cls = bg_pre + (size_t)4 * g.Dout * g.F * g.Cout + (size_t)d * 9 * g.Cout;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int gy = ty0 + 2 * wv + (row >> 4), gx = tx0 + (row & 15);
            const int q = 3 * (gy == 0 ? 0 : (gy >= g.H - 1 ? 2 : 1)) + (gx == 0 ? 0 : (gx >= g.W - 1 ? 2 : 1));
            acc0[r] = bg_cls[q * g.Cout + n0];
            acc1[r] = bg_cls[q * g.Cout + n1];
        }
    }
    float skip0 = 0.f, skip1 = 0.f;            // constants of the depth taps that were not executed
    if (skipped && active) {
        const float *bg_tap = bg_pre + (size_t)g.Dout * g.F * g.Cout + (size_t)d * 3 * g.Cout;
#pragma unroll
        for (int kd = 0; kd < 3; ++kd)
            if ((skipped >> kd) & 1u) { skip0 += bg_tap[kd * g.Cout + n0]; skip1 += bg_tap[kd * g.Cout + n1]; }
    }
    float bgv0 = (bg_pre ? bg_pre[(size_t)d * g.Cout + n0] : 0.f) + bias0, bgv1 = (bg_pre ? bg_pre[(size_t)d * g.Cout + n1] : 0.f) + bias1;
    if (relu) { bgv0 = fmaxf(bgv0, 0.f); bgv1 = fmaxf(bgv1, 0.f); }
    // site-mask bytes fetched up front (see conv3d.hip: a load between the stores made every store pair wait)
    unsigned site_on = active ? 0xffffu : 0u;
    if (out_mask && active) {
        unsigned char mk[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int gy = min(ty0 + 2 * wv + (row >> 4), g.H - 1), gx = min(tx0 + (row & 15), g.W - 1);
            mk[r] = out_mask[((size_t)d * g.H + gy) * g.W + gx];
        }
        site_on = 0u;
#pragma unroll
        for (int r = 0; r < 16; ++r) site_on |= (mk[r] ? 1u : 0u) << r;
    }
    float s1a = 0.f, s2a = 0.f, s1b = 0.f, s2b = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int gy = ty0 + 2 * wv + (row >> 4), gx = tx0 + (row & 15);
        float v0 = (acc0[r] + skip0) + bias0, v1 = (acc1[r] + skip1) + bias1;
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


// ------------------------------------------------------------------------------------------
// The same gather with a 16 x 16-site patch per workgroup and 64 sites x 64 channels per wave (four accumulator tiles).
//
// At 32 cycles per bf16 MFMA a stage of the 8 x 16 kernel lasts 3,456 matrix cycles per wave, during which the workgroup
// pulls 73 KB of (pre-split) weights and 23 KB of halo through L2: with two workgroups on each of 256 CUs that is ~17 TB/s
// of L2 reads -- half the aggregate L2 peak -- and one ds_read_b128 per MFMA.  Twice the sites per workgroup halve the weight
// bytes and the weight-operand reads per MFMA (12 MFMAs per 8 operand fragments instead of 6 per 6) and put twice the
// matrix work between two barriers.  The activity flags stay per 8 x 16 tile (what activity.hip produces): a workgroup
// covers the tiles (tx, 2 ty) and (tx, 2 ty + 1) and ORs their flags -- computing a background half is exact, just not
// needed.  Used when the launch has enough of these larger units to fill the GPU (launch_gather_split).
// ------------------------------------------------------------------------------------------
constexpr int TH2 = 16, HH2 = TH2 + 2;

__global__ __launch_bounds__(256, 2) void conv3d_gather_split16(const float *__restrict__ in,
                                                                const unsigned short *__restrict__ wsp,
                                                                const float *__restrict__ bias,
                                                                float *__restrict__ out, double *__restrict__ stats,
                                                                Geom g, int relu, const int *__restrict__ in_hflag,
                                                                const unsigned char *__restrict__ out_mask,
                                                                const float *__restrict__ bg_pre, int border_active,
                                                                const int *__restrict__ only_tiles,
                                                                unsigned long long *__restrict__ exec_stages) {
    __shared__ __attribute__((aligned(16))) unsigned char s_halo[HH2 * HW * ROWB];
    __shared__ __attribute__((aligned(16))) unsigned char s_w[3][BN * ROWB];
    __shared__ float s_red[4][2 * BN];
    const int tiles_x = (g.W + TW - 1) / TW, tiles_y8 = (g.H + TH - 1) / TH;
    const int ntiles8 = tiles_x * tiles_y8;
    const int tx = blockIdx.x % tiles_x, ty16 = blockIdx.x / tiles_x;
    const int tx0 = tx * TW, ty0 = ty16 * TH2;
    const int t_top = (2 * ty16) * tiles_x + tx;
    const bool has_bot = 2 * ty16 + 1 < tiles_y8;
    const int t_bot = has_bot ? t_top + tiles_x : t_top;
    const int d = blockIdx.y, nb = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int nchunks = g.Cin / BK;
    if (only_tiles && !(only_tiles[(size_t)d * ntiles8 + t_top] | only_tiles[(size_t)d * ntiles8 + t_bot])) return;
    const bool on_border = tx0 == 0 || ty0 == 0 || tx0 + TW >= g.W || ty0 + TH2 >= g.H;
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

    f32x16 acc[2][2];                                   // [site tile m: rows 4 wv + 2 m, + 1][channel tile: n0, n0 + 32]
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    int a_base[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) a_base[m] = ((4 * wv + 2 * m + (li >> 4)) * HW + (li & 15)) * ROWB + lh * 16;
    const int b_base = li * ROWB + lh * 16;

    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 wreg[6];
    auto load_w3 = [&](int kd, int a, int cc) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int c = tid + 256 * u;
            const int t = c >> 9, rem = c & 511;
            const unsigned char *tile = (const unsigned char *)wsp +
                ((((size_t)kd * 9 + a * 3 + t) * nchunks + cc) * g.Cout + (size_t)nb * BN) * (2 * BK * 2);
            wreg[u] = *(const u32x4 *)(tile + (size_t)rem * 16);
        }
    };
    auto store_w3 = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int c = tid + 256 * u;
            const int t = c >> 9, rem = c & 511;
            const int n = rem >> 3, piece = rem & 7;
            *(u32x4 *)(s_w[t] + n * ROWB + piece * 16) = wreg[u];
        }
    };
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
    // executed work in units of the 8 x 16-tile stage (what the roofline prices): this workgroup covers one or two of them
    if (exec_stages && nstages > 0 && threadIdx.x == 0) atomicAdd(exec_stages, (unsigned long long)nstages * (has_bot ? 2 : 1));
    auto stage_of = [&](int st, int &kd, int &ds, int &cc) __attribute__((always_inline)) {
        const int i = st / nchunks;
        cc = st - i * nchunks;
        kd = i == 0 ? kd_l[0] : (i == 1 ? kd_l[1] : kd_l[2]);
        ds = i == 0 ? ds_l[0] : (i == 1 ? ds_l[1] : ds_l[2]);
    };
    constexpr int NH = (HH2 * HW * 8 + 255) / 256;      // 16-byte halo pieces per thread (11)
    int h_off[NH], h_lds[NH];
#pragma unroll
    for (int u = 0; u < NH; ++u) {
        const int c = tid + 256 * u;
        h_off[u] = -1;
        h_lds[u] = -1;
        if (c < HH2 * HW * 8) {
            const int r = c >> 3, part = c & 7;
            const int gy = ty0 - 1 + r / HW, gx = tx0 - 1 + r % HW;
            h_lds[u] = r * ROWB + part * 8;
            if (gy >= 0 && gy < g.H && gx >= 0 && gx < g.W) h_off[u] = (gy * g.W + gx) * g.Cin + part * 4;
        }
    }
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 hreg[NH];
    auto load_halo = [&](int st) __attribute__((always_inline)) {
        int kd, ds, cc;
        stage_of(st, kd, ds, cc);
        const float *img = in + (size_t)ds * g.H * g.W * g.Cin + cc * BK;
#pragma unroll
        for (int u = 0; u < NH; ++u)
            hreg[u] = h_off[u] >= 0 ? *(const f32x4 *)(img + h_off[u]) : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto store_halo = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < NH; ++u)
            if (h_lds[u] >= 0) {
                uint2 hi, lo;
                split4(make_float4(hreg[u][0], hreg[u][1], hreg[u][2], hreg[u][3]), &hi, &lo);
                *(uint2 *)(s_halo + h_lds[u]) = hi;
                *(uint2 *)(s_halo + h_lds[u] + 64) = lo;
            }
    };
    auto load_wrow = [&](int st, int a) __attribute__((always_inline)) {
        int kd, ds, cc;
        stage_of(st, kd, ds, cc);
        load_w3(kd, a, cc);
    };
    auto compute_row = [&](int a) __attribute__((always_inline)) {
#pragma unroll
        for (int b = 0; b < 3; ++b) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 b0h = __builtin_bit_cast(bf16x8, *(const uint4 *)(s_w[b] + b_base + s2 * 32));
                const bf16x8 b0l = __builtin_bit_cast(bf16x8, *(const uint4 *)(s_w[b] + b_base + 64 + s2 * 32));
                const bf16x8 b1h = __builtin_bit_cast(bf16x8, *(const uint4 *)(s_w[b] + b_base + 32 * ROWB + s2 * 32));
                const bf16x8 b1l = __builtin_bit_cast(bf16x8, *(const uint4 *)(s_w[b] + b_base + 32 * ROWB + 64 + s2 * 32));
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int a_off = a_base[m] + (a * HW + b) * ROWB;
                    const bf16x8 ah = __builtin_bit_cast(bf16x8, *(const uint4 *)(s_halo + a_off + s2 * 32));
                    const bf16x8 al = __builtin_bit_cast(bf16x8, *(const uint4 *)(s_halo + a_off + 64 + s2 * 32));
                    acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b0h, acc[m][0], 0, 0, 0);
                    acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b1h, acc[m][1], 0, 0, 0);
                    acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b0l, acc[m][0], 0, 0, 0);
                    acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b1l, acc[m][1], 0, 0, 0);
                    acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b0h, acc[m][0], 0, 0, 0);
                    acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b1h, acc[m][1], 0, 0, 0);
                }
            }
        }
    };
    if (nstages > 0) {
        load_wrow(0, 0);
        load_halo(0);
    }
    for (int st = 0; st < nstages; ++st) {
        const int nxt = st + 1 < nstages ? st + 1 : st;
        __syncthreads();
        store_halo();
        store_w3();
        __syncthreads();
        load_wrow(st, 1);
        load_halo(nxt);
        compute_row(0);
        __syncthreads();
        store_w3();
        __syncthreads();
        load_wrow(st, 2);
        compute_row(1);
        __syncthreads();
        store_w3();
        __syncthreads();
        load_wrow(nxt, 0);
        compute_row(2);
    }

    // ---- epilogue (same arithmetic per site as conv3d_gather_split)
    const int n0 = nb * BN + li, n1 = n0 + 32;
    const float bias0 = bias ? bias[n0] : 0.f, bias1 = bias ? bias[n1] : 0.f;
    float skip0 = 0.f, skip1 = 0.f;
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
    for (int m = 0; m < 2; ++m) {
        if (idle_border) {
            const float *bg_cls = bg_pre + (size_t)4 * g.Dout * g.F * g.Cout + (size_t)d * 9 * g.Cout;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int gy = ty0 + 4 * wv + 2 * m + (row >> 4), gx = tx0 + (row & 15);
                const int q = 3 * (gy == 0 ? 0 : (gy >= g.H - 1 ? 2 : 1)) + (gx == 0 ? 0 : (gx >= g.W - 1 ? 2 : 1));
                acc[m][0][r] = bg_cls[q * g.Cout + n0];
                acc[m][1][r] = bg_cls[q * g.Cout + n1];
            }
        }
        unsigned site_on = active ? 0xffffu : 0u;
        if (out_mask && active) {
            unsigned char mk[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int gy = min(ty0 + 4 * wv + 2 * m + (row >> 4), g.H - 1), gx = min(tx0 + (row & 15), g.W - 1);
                mk[r] = out_mask[((size_t)d * g.H + gy) * g.W + gx];
            }
            site_on = 0u;
#pragma unroll
            for (int r = 0; r < 16; ++r) site_on |= (mk[r] ? 1u : 0u) << r;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int gy = ty0 + 4 * wv + 2 * m + (row >> 4), gx = tx0 + (row & 15);
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
            double *fstats = stats + (size_t)(d / g.Dout) * MVX_REP * 2 * g.Cout;
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
    return MVX_EINVAL;
}

static void launch_gather_split(hipStream_t st, int planes, int nblocks, const float *in, const unsigned short *wsp,
                                const float *bias, float *out, double *stats, const Geom &g, int relu, const int *in_hflag,
                                const unsigned char *out_mask, const float *bg_pre, int border_active, const int *only_tiles,
                                unsigned long long *exec_stages) {
    const int tiles_x = (int)mvx_cdiv(g.W, TW);
    const long long units16 = (long long)tiles_x * mvx_cdiv(g.H, TH2) * planes * nblocks;
    if (units16 >= g_split16_min_units)
        hipLaunchKernelGGL(conv3d_gather_split16, dim3(tiles_x * mvx_cdiv(g.H, TH2), planes, nblocks), dim3(256), 0, st, in, wsp,
                           bias, out, stats, g, relu, in_hflag, out_mask, bg_pre, border_active, only_tiles, exec_stages);
    else
        hipLaunchKernelGGL(conv3d_gather_split, dim3(tiles_x * mvx_cdiv(g.H, TH), planes, nblocks), dim3(256), 0, st, in, wsp, bias,
                           out, stats, g, relu, in_hflag, out_mask, bg_pre, border_active, only_tiles, exec_stages);
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

__global__ __launch_bounds__(WG_THREADS) void conv3d_wgrad_split(const float *__restrict__ in,
                                                                 const float *__restrict__ dz,
                                                                 float *__restrict__ slabs, Geom g,
                                                                 int tiles_per_strip, const int *__restrict__ step_list,
                                                                 const int *__restrict__ step_count,
                                                                 const float *__restrict__ c_in) {
    __shared__ __attribute__((aligned(16))) unsigned short s_xh[HH * HW][BK], s_xl[HH * HW][BK];
    __shared__ __attribute__((aligned(16))) unsigned short s_zh[2][TH * TW][32], s_zl[2][TH * TW][32];
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
                uint2 hi, lo;
                split4(xr[u], &hi, &lo);
                *(uint2 *)(&s_xh[c >> 3][(c & 7) * 4]) = hi;
                *(uint2 *)(&s_xl[c >> 3][(c & 7) * 4]) = lo;
            }
        }
#pragma unroll
        for (int u = 0; u < NZ; ++u) {
            const int c = tid + WG_THREADS * u;
            if (c < TH * TW * 16) {
                const int r = c >> 4, part = c & 15;
                uint2 hi, lo;
                split4(zr[u], &hi, &lo);
                *(uint2 *)(&s_zh[part >> 3][r][(part & 7) * 4]) = hi;
                *(uint2 *)(&s_zl[part >> 3][r][(part & 7) * 4]) = lo;
            }
        }
        __syncthreads();
        const int nxt = next_live(cur + 1);
        if (nxt < nsteps) load_step(nxt);
#pragma unroll 2
        for (int ks = 0; ks < TH; ++ks) {                 // 16 sites (one patch row) per MFMA k-step
            const int hr0 = (ks + ta) * HW + tb + kbase + q, hr1 = hr0 + 4;      // halo rows of sites kbase+q, +4
            const int zr0 = ks * TW + kbase + q, zr1 = zr0 + 4;
            const bf16x8 ah = tr_frag(&s_xh[hr0][pcol], &s_xh[hr1][pcol]);
            const bf16x8 al = tr_frag(&s_xl[hr0][pcol], &s_xl[hr1][pcol]);
            const bf16x8 b0h = tr_frag(&s_zh[0][zr0][pcol], &s_zh[0][zr1][pcol]);
            const bf16x8 b0l = tr_frag(&s_zl[0][zr0][pcol], &s_zl[0][zr1][pcol]);
            const bf16x8 b1h = tr_frag(&s_zh[1][zr0][pcol], &s_zh[1][zr1][pcol]);
            const bf16x8 b1l = tr_frag(&s_zl[1][zr0][pcol], &s_zl[1][zr1][pcol]);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b0h, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b1h, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b0l, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b1l, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b0h, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b1h, acc1, 0, 0, 0);
        }
        cur = nxt;
    }
    float *o = slabs + ((((size_t)strip * 3 + kd) * 9 + tap) * g.Cin + cc * BK) * BN;
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
    if (din <= 0 || dout <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0) return MVX_EINVAL;
    if (sd < 1 || sd > 2 || pd < 0 || pd > 1) return MVX_EINVAL;
    if (cin % BK || cout % BN) return MVX_ESIZE;
    return MVX_OK;
}

}  // namespace

extern "C" int mvx_conv3d_pack_weights_split(const float *w, void *wsplit, int32_t cout, int32_t cin, int32_t for_dgrad,
                                             void *stream) {
    MVX_CHECK_ARG(w && wsplit && cout > 0 && cin > 0);
    MVX_CHECK_ARG((for_dgrad ? cout : cin) % BK == 0);
    const long long total = 27ll * cout * cin;
    hipLaunchKernelGGL(pack_weights_split, dim3(mvx_cdiv(total, 256) > 2048 ? 2048 : mvx_cdiv(total, 256)), dim3(256), 0,
                       (hipStream_t)stream, w, (unsigned short *)wsplit, cout, cin, for_dgrad);
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
    launch_gather_split(st, dout, cout / BN, in, (const unsigned short *)wsplit, bias, out, stats, g, relu, nullptr, nullptr, nullptr,
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
    launch_gather_split(st, dout * n_frames, cout / BN, in, (const unsigned short *)wsplit, bias, out, stats, g,
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
                              int32_t w, int32_t cin, int32_t cout, int32_t stride_d, int32_t pad_d,
                              const int32_t *only_tiles, uint64_t *exec_stages, int32_t n_frames, void *stream) {
    MVX_CHECK_ARG(dz && wsplit_dgrad && dx);
    MVX_CHECK_ARG(n_frames >= 1 && n_frames <= MVX_MAX_FRAMES);
    int rc = check_geom(din, dout, h, w, cout, cin, stride_d, pad_d);
    if (rc) return rc;
    Geom g{dout, din, h, w, cout, cin, stride_d, pad_d, 1, n_frames};
    launch_gather_split((hipStream_t)stream, din * n_frames, cin / BN, dz, (const unsigned short *)wsplit_dgrad, nullptr, dx, nullptr,
                        g, 0, nullptr, nullptr, nullptr, 0, only_tiles, (unsigned long long *)exec_stages);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" int mvx_conv3d_dgrad_tiles_split_frames(const float *dz, const void *wsplit_dgrad, float *dx, int32_t din,
                                                   int32_t dout, int32_t h, int32_t w, int32_t cin, int32_t cout,
                                                   int32_t stride_d, int32_t pad_d, const int32_t *dx_tile_flags,
                                                   uint64_t *exec_stages, int32_t n_frames, void *stream) {
    MVX_CHECK_ARG(dx_tile_flags);
    return launch_dgrad_split(dz, wsplit_dgrad, dx, din, dout, h, w, cin, cout, stride_d, pad_d, dx_tile_flags, exec_stages,
                              n_frames, stream);
}

extern "C" int mvx_conv3d_dgrad_tiles_split(const float *dz, const void *wsplit_dgrad, float *dx, int32_t din, int32_t dout,
                                            int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t stride_d,
                                            int32_t pad_d, const int32_t *dx_tile_flags, void *stream) {
    MVX_CHECK_ARG(dx_tile_flags);
    return launch_dgrad_split(dz, wsplit_dgrad, dx, din, dout, h, w, cin, cout, stride_d, pad_d, dx_tile_flags, nullptr, 1, stream);
}

extern "C" int mvx_conv3d_dgrad_split(const float *dz, const void *wsplit_dgrad, float *dx, int32_t din, int32_t dout,
                                      int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t stride_d, int32_t pad_d,
                                      void *stream) {
    return launch_dgrad_split(dz, wsplit_dgrad, dx, din, dout, h, w, cin, cout, stride_d, pad_d, nullptr, nullptr, 1, stream);
}

extern "C" int mvx_conv3d_wgrad_split(const float *in, const float *dz, float *dw, int32_t din, int32_t dout, int32_t h,
                                      int32_t w, int32_t cin, int32_t cout, int32_t stride_d, int32_t pad_d,
                                      int32_t flags, void *workspace, size_t workspace_bytes, void *stream) {
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
    hipLaunchKernelGGL(conv3d_wgrad_split, dim3(nstrips, 3 * (cin / BK)), dim3(WG_THREADS), 0, st, in, dz,
                       (float *)workspace, g, per, (const int *)nullptr, (const int *)nullptr, (const float *)nullptr);
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
    hipLaunchKernelGGL(conv3d_wgrad_split, dim3(nstrips, 3 * (cin / BK)), dim3(WG_THREADS), 0, st, in, dz, slabs, g, 0,
                       (const int *)list, (const int *)count, c_in);
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
    launch_gather_split(st, n_frames, cout / BN, in, (const unsigned short *)wsplit, bias, out, stats, g, flags & MVX_FLAG_RELU,
                        nullptr, nullptr, nullptr, 0, nullptr, nullptr);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" int mvx_conv2d_dgrad_split_frames(const float *dz, const void *wsplit_dgrad, float *dx, int32_t h, int32_t w,
                                             int32_t cin, int32_t cout, int32_t n_frames, void *stream) {
    return launch_dgrad_split(dz, wsplit_dgrad, dx, 1, 1, h, w, cin, cout, 1, 1, nullptr, nullptr, n_frames, stream);
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
                                             int32_t cout, void *workspace, size_t workspace_bytes, int32_t n_frames,
                                             void *stream) {
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
        hipLaunchKernelGGL(conv3d_wgrad_split, dim3(nstrips, 3 * (cin / BK)), dim3(WG_THREADS), 0, st, in, dz + (size_t)nb * BN,
                           slabs, g, 0, (const int *)list, (const int *)count, (const float *)nullptr);
        MVX_LAUNCH_CHECK();
        hipLaunchKernelGGL(wgrad_reduce_split, dim3(mvx_cdiv(per_slab, 256)), dim3(256), 0, st, (const float *)slabs,
                           dw3 + (size_t)nb * BN * cin * 27, nstrips, cin, 0);
        MVX_LAUNCH_CHECK();
    }
    return MVX_OK;
}
