// Dense 3-D convolution (3x3x3, stride/pad in depth only) as an implicit GEMM on the CDNA4
// matrix cores, channels-last.  Stands for the three cuDNN Conv3d calls of the reference's
// CML (modules/voxelnet/Pipe.py:31-43, modules/layers/Blocks.py:20-29) and their autograd
// (dgrad / wgrad).
//
// Layout: activations [D][H][W][C] f32 (one frame), C contiguous -> one filter tap of one
// site is a contiguous run of C floats.  fp32 parity mode uses v_mfma_f32_32x32x2_f32
// (exact f32 fma chain, 64 FLOP/clk/SIMD).
//
// Gather kernel (forward and dgrad):  out[d][y][x][n] = sum_{kd,a,b,c} in[src(d,kd)][y+a-1][x+b-1][c]
//                                                       * wpk[kd][a][b][c][n]
//   - one workgroup (4 waves) owns an 8x16 patch of output sites of one depth plane and a
//     block of 64 output channels; wave w owns rows 2w,2w+1 of the patch (32 sites) x 64 n;
//   - K loop = (kd) x (32-channel chunk) x (9 in-plane taps): the 10x18-site halo of the
//     patch is staged in LDS once per (kd, chunk) and all 9 taps read it in place; the
//     64x32 weight tile of each tap is prefetched into registers under the previous tap's MFMAs;
//   - LDS rows are padded to 36 floats so the ds_read_b128 operand reads (4 consecutive k per
//     lane, k order permuted identically for A and B) spread over all 64 banks;
//   - epilogue: + bias, ReLU, per-channel sum / sum-of-squares for the following BatchNorm
//     (block partials -> f64 atomics), 128-byte row-segment stores.
//
// Wgrad kernel:  dW[kd][a][b][c][n] = sum_{d,y,x} in[src(d,kd)][y+a-1][x+b-1][c] * dz[d][y][x][n]
//   - one workgroup of 9 waves per (strip of patches, kd, 32-channel chunk); wave t owns tap t
//     and a 32(c) x 64(n) accumulator; the reduction runs over sites; per-strip partial slabs
//     are summed by a second, deterministic kernel (no float atomics).
#include "common.h"
#include "split_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// In-kernel time stamps of ONE unit of the gather kernel (tools/probes/gather_stamps.hip compiles this file with
// -DMVX_GATHER_STAMPS; the library build has none of it): s_memtime read by thread 0 right after a barrier, where
// lgkmcnt is zero anyway.
#ifdef MVX_GATHER_STAMPS
__device__ unsigned long long g_stamps[256];
__device__ int g_stamp_unit[3] = {0, 0, 0};            // (tile, plane, channel block) of the stamped unit
__device__ volatile int g_stamp_wg = -1;               // workgroup that ran the stamped unit: its NEXT unit stamps 201 / 202
#define MVX_STAMP(k)                                                                                       \
    do {                                                                                                   \
        if (stamped && threadIdx.x == 0 && (k) < 256) g_stamps[(k)] = __builtin_amdgcn_s_memtime();        \
    } while (0)
#else
#define MVX_STAMP(k) do { } while (0)
#endif

constexpr int TH = 8, TW = 16;          // output patch of a workgroup
constexpr int HH = TH + 2, HW = TW + 2; // halo
constexpr int BK = 32;                  // channels per K chunk
constexpr int PITCH = BK + 4;           // LDS row pitch in floats (bank-conflict padding)
constexpr int BN = 64;                  // output channels per workgroup

struct Geom {
    int Din, Dout, H, W, Cin, Cout;     // gather view: in has Cin channels, out has Cout; Din / Dout = planes PER FRAME
    int sd, pd;                         // depth stride / padding of the FORWARD conv
    int mode;                           // 0 forward gather, 1 dgrad gather
    int F = 1;                          // frames stacked along the depth axis: global plane = frame * planes + local plane
    int tap_lo = 0, tap_hi = 3;         // in-plane taps (rows AND columns) [tap_lo, tap_hi) carry weight; the others are skipped
                                        // (stride-2 convolutions evaluated on the space-to-depth image use a 2x2 window)
    int s2d = 0;                        // > 0: the 2x2 window stands for a stride-2 3x3 kernel on the space-to-depth image whose
                                        // channels are four parity blocks [pr][pc] of s2d channels each: window tap (ta, tb)
                                        // carries weight for parity (pr, pc) only if (ta == 1 || pr == 1) && (tb == 1 || pc == 1)
                                        // (include/mvx_hip.h, MVX_FLAG_TAPS2) -- 9 of the 16 (tap, parity) blocks; the others are
                                        // structural zeros and are not executed
};

// valid window taps of parity block p = pr * 2 + pc as a 4-bit mask, bit (ta * 2 + tb) (see Geom::s2d)
__device__ __forceinline__ unsigned s2d_tap_mask(int p) {
    const int pr = p >> 1, pc = p & 1;
    unsigned m = 8u;                                   // (1,1) always
    if (pr) m |= 2u;                                   // (0,1)
    if (pc) m |= 4u;                                   // (1,0)
    if (pr && pc) m |= 1u;                             // (0,0)
    return m;
}

// source depth plane (global) of GLOBAL output plane d for depth tap kd; -1 if the tap falls outside the frame's volume
__device__ __forceinline__ int src_depth(const Geom &g, int d, int kd) {
    if (g.mode == 0) return mvx_src_plane(d, g.Din, g.Dout, g.sd, g.pd, kd);
    // dgrad gather: the result (dx) has g.Dout planes per frame, the source (dz) g.Din
    return mvx_dst_plane(d, g.Dout, g.Din, g.sd, g.pd, kd);
}

// ------------------------------------------------------------------------------------------
// weight packing: torch layout W[co][ci][kd][kh][kw] -> wpk[kd][a][b][chunk][n][BK]
//   forward: n = co, k = ci, (a,b) = (kh,kw)
//   dgrad  : n = ci, k = co, (a,b) = (2-kh, 2-kw)
// ------------------------------------------------------------------------------------------
__global__ void pack_weights(const float *__restrict__ w, float *__restrict__ wpk, int Co, int Ci, int dgrad, int src2d) {
    const int K = dgrad ? Co : Ci, N = dgrad ? Ci : Co;
    const int nch = K / BK;
    const long long total = 27ll * K * N;
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
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
        // src2d: the source is a 2-D kernel W[co][ci][3][3] standing for the middle depth slice (depth taps 0 and 2
        // are zero and, with din = dout = 1 and pad_d = 1, never executed)
        if (src2d) wpk[e] = kd == 1 ? w[(((long long)co * Ci + ci) * 3 + kh) * 3 + kw] : 0.f;
        else wpk[e] = w[((((long long)co * Ci + ci) * 3 + kd) * 3 + kh) * 3 + kw];
    }
}

// ------------------------------------------------------------------------------------------
// gather convolution (forward and dgrad), software-pipelined
//
// Co-resident workgroups start together and share the
// MFMA pipe, which keeps them in lock step, so a halo fetch issued at the top of a (depth tap, chunk)
// stage is exposed in BOTH of them at once.  Here the halo of stage s+1 is fetched into registers
// right after stage s starts computing, and the weights go to LDS a tap ROW (3 taps) at a time, with
// the next row in flight under 96 MFMAs: 6 barriers per stage instead of 19, no exposed global latency.
// Load order matters (vector-memory returns are in order): the weight row needed next is always issued
// BEFORE the long-latency halo fetch, so waiting for it does not wait for the halo.
// ------------------------------------------------------------------------------------------
constexpr int WROW = BN * BK;              // one tap's weight tile in LDS: [64 cout][32 k], rows UNPADDED, the eight 16-byte
                                           // slots of a row XOR-swizzled with (cout >> 1) & 7 -- conflict-free for the
                                           // ds_read_b128 lane groups like the padded pitch was, and 3 KB smaller, which
                                           // (with the statistics scratch aliased onto it) lets THREE workgroups share a CU
// Halo row stride: a multiple of 64 floats, so that the two patch rows a wave reads (lanes 0-15 / 16-31)
// start on the same 16-byte slot of the 256-byte bank row.  ds_read_b128 is served in the lane groups
// {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32): with site pitch 36 floats (9 slots, odd) the slots of one
// row are a permutation of 0..15, and equal row phases make every group hit 16 distinct slots.
constexpr int HROW = ((HW * PITCH + 63) / 64) * 64;

// Launch geometry: (tiles, output planes, 64-channel blocks).  An XCD-contiguous remap of the (tile, plane)
// space was measured and was not faster (forward equal, stride-2 dgrad slower), so the plain grid stays.
inline dim3 gather_grid(const Geom &g) { return dim3(mvx_cdiv(g.W, TW) * mvx_cdiv(g.H, TH), g.Dout * g.F, g.Cout / BN); }

// One unit of work = (tile, output plane d, 64-channel block nb); `ntiles` tiles per plane.
// TLO / THI: in-plane taps [TLO, THI) (rows and columns) carry weight -- compile-time, so that the two-row pipeline
// of the 2 x 2 window keeps its prefetch registers in VGPRs (a run-time choice between two pipelines demoted them
// to scratch).
// NB2: 32-channel accumulator tiles per wave -- 2: the unit covers 64 output channels (BN); 1: 32 (the "narrow" unit, twice
// the units of half the work each: launches too small to fill the GPU with 64-channel units, see launch_gather).
template <int TLO, int THI, int NB2 = 2>
__device__ __forceinline__ void gather_unit(const int tile, const int d, const int nb, const int ntiles,
                                            float *__restrict__ s_halo, float *__restrict__ s_w, double (*s_red)[2 * BN],
                                            const float *__restrict__ in, const float *__restrict__ wpk,
                                            const float *__restrict__ bias, float *__restrict__ out,
                                            double *__restrict__ stats, const Geom &g, int relu,
                                            const int *__restrict__ in_hflag, const unsigned char *__restrict__ out_mask,
                                            const float *__restrict__ bg_pre, int border_active,
                                            unsigned long long *__restrict__ exec_stages,
                                            const int *__restrict__ only_tiles) {
    const int tiles_x = (g.W + TW - 1) / TW;
    const int tx0 = (tile % tiles_x) * TW, ty0 = (tile / tiles_x) * TH;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int nchunks = g.Cin / BK;
    constexpr int BNW = 32 * NB2;                    // output channels of this unit
    // restricted launch: only the flagged output tiles are produced, the others are left untouched
    if (only_tiles && !only_tiles[(size_t)d * ntiles + tile]) return;

    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }

    const int my_ty = 2 * wv + (li >> 4), my_tx = li & 15;
    const int a_base = my_ty * HROW + my_tx * PITCH + 4 * lh;
    // weight rows li and 32 + li share the swizzle key; slot of k-group q: ((2q | lh) ^ key)
    const int wkey = (li >> 1) & 7;
    int b_off[BK / 8];
#pragma unroll
    for (int q = 0; q < BK / 8; ++q) b_off[q] = li * BK + 4 * (((2 * q) | lh) ^ wkey);

    // valid depth taps of this output plane (block-uniform), packed as (kd, source plane) pairs.
    // border_active bit 0: tiles on the image border always count as active (zero padding is not the background);
    // bit 1: bg_pre is followed by per-depth-tap constants bg_tap[plane][kd][cout] -- in an INTERIOR tile a depth tap whose
    // source halo holds no active site contributes exactly that constant (every source site is the plane's background
    // value and the window stays inside the image), so it is not executed: `skipped` collects those taps for the epilogue.
    const bool on_border = tx0 == 0 || ty0 == 0 || tx0 + TW >= g.W || ty0 + TH >= g.H;
    const bool skip_taps = in_hflag && (border_active & 2) && !on_border;
    int kd_l[3] = {0, 0, 0}, ds_l[3] = {0, 0, 0}, nk = 0;
    unsigned skipped = 0;
    int any_flag = 0;
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) {
        const int ds = src_depth(g, d, kd);
        if (ds >= 0) {
            const int fl = in_hflag ? in_hflag[(size_t)ds * ntiles + tile] : 1;
            any_flag |= fl;
            if (skip_taps && !fl) { skipped |= 1u << kd; continue; }
            if (nk == 0) { kd_l[0] = kd; ds_l[0] = ds; }
            else if (nk == 1) { kd_l[1] = kd; ds_l[1] = ds; }
            else { kd_l[2] = kd; ds_l[2] = ds; }
            ++nk;
        }
    }
    // ... and a BORDER tile without any active source site is not convolved either: its sites are the plane's background
    // inside and, on the outermost ring, the background minus the taps that fall into the zero padding -- nine
    // per-(plane, channel) constants by position class (bg_cls, after bg_tap in the buffer).
    const bool idle_border = in_hflag && (border_active & 2) && on_border && !any_flag;
    if (idle_border) nk = 0;
    const int nstages = nk * nchunks;

    // Background tiles (see activity.hip): no non-background source site in the halo of any depth tap and the
    // window never leaves the image -> every output site of the tile is the per-plane constant.
    bool active = true;
    if (in_hflag) active = (((border_active & 1) && on_border) || any_flag) != 0;
    // executed work only, in 64-channel stages: the two narrow units of a channel pair run the same stages (activity
    // does not depend on nb), so the even one reports for both
    if (exec_stages && active && threadIdx.x == 0 && (NB2 == 2 || !(nb & 1))) atomicAdd(exec_stages, (unsigned long long)nstages);
    if (active && nstages > 0) {
    auto stage_kd = [&](int st, int &kd, int &ds, int &cc) __attribute__((always_inline)) {
        const int i = st / nchunks;
        cc = st - i * nchunks;
        kd = i == 0 ? kd_l[0] : (i == 1 ? kd_l[1] : kd_l[2]);
        ds = i == 0 ? ds_l[0] : (i == 1 ? ds_l[1] : ds_l[2]);
    };

    // per-thread halo slots: site r = c >> 3, 16-byte part c & 7, c = tid + 256 u
    int h_off[6];               // global float offset inside a (plane, chunk) image, or -1 (outside -> zeros)
    int h_lds[6];               // LDS float offset of the slot, or -1 (no slot: 1440 slots over 1536 threads x u)
#pragma unroll
    for (int u = 0; u < 6; ++u) {
        const int c = tid + 256 * u;
        h_off[u] = -1;
        h_lds[u] = -1;
        if (c < HH * HW * 8) {
            const int r = c >> 3, part = c & 7;
            const int ry = r / HW, rx = r - ry * HW;
            const int gy = ty0 - 1 + ry, gx = tx0 - 1 + rx;
            h_lds[u] = ry * HROW + rx * PITCH + part * 4;
            if (gy >= 0 && gy < g.H && gx >= 0 && gx < g.W) h_off[u] = (gy * g.W + gx) * g.Cin + part * 4;
        }
    }
    f32x4 hreg[6], wreg[6];     // native vectors: HIP's float4 struct copies become memcpy and pin the arrays in scratch
    auto load_halo = [&](int st) __attribute__((always_inline)) {
        int kd, ds, cc;
        stage_kd(st, kd, ds, cc);
        const float *img = in + (size_t)ds * g.H * g.W * g.Cin + cc * BK;
#pragma unroll
        for (int u = 0; u < 6; ++u)
            hreg[u] = h_off[u] >= 0 ? *(const f32x4 *)(img + h_off[u]) : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto store_halo = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 6; ++u)
            if (h_lds[u] >= 0) *(f32x4 *)(s_halo + h_lds[u]) = hreg[u];
    };
    auto load_wrow = [&](int st, int row) __attribute__((always_inline)) {
        int kd, ds, cc;
        stage_kd(st, kd, ds, cc);
        const float *row0 = wpk + ((((size_t)kd * 9 + row * 3) * nchunks + cc) * g.Cout + (size_t)nb * BNW) * BK + (size_t)tid * 4;
        const size_t tap_stride = (size_t)nchunks * g.Cout * BK;
        if (NB2 == 2) {
#pragma unroll
            for (int v = 0; v < 6; ++v) wreg[v] = *(const f32x4 *)(row0 + (v >> 1) * tap_stride + (v & 1) * 1024);
        } else {                                     // 32 x 32 floats per tap: one 16-byte piece per thread
#pragma unroll
            for (int v = 0; v < 3; ++v) wreg[v] = *(const f32x4 *)(row0 + v * tap_stride);
        }
    };
    auto store_wrow = [&]() __attribute__((always_inline)) {
        if (NB2 == 2) {
#pragma unroll
            for (int v = 0; v < 6; ++v) {
                const int c = tid + 256 * (v & 1);
                *(f32x4 *)(s_w + (v >> 1) * WROW + (c >> 3) * BK + (((c & 7) ^ ((c >> 4) & 7)) * 4)) = wreg[v];
            }
        } else {
#pragma unroll
            for (int v = 0; v < 3; ++v)
                *(f32x4 *)(s_w + v * WROW + (tid >> 3) * BK + (((tid & 7) ^ ((tid >> 4) & 7)) * 4)) = wreg[v];
        }
    };
    auto compute_row = [&](int row, unsigned colmask = 7u) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            if (t < TLO || t >= THI) continue;
            if (!((colmask >> t) & 1u)) continue;     // block-uniform: a structurally zero tap of the space-to-depth form
            const int a_off = a_base + row * HROW + t * PITCH;
#pragma unroll
            for (int q = 0; q < BK / 8; ++q) {
                const float4 av = *(const float4 *)(s_halo + a_off + 8 * q);
                const float4 b0 = *(const float4 *)(s_w + t * WROW + b_off[q]);
                if (NB2 == 2) {
                    const float4 b1 = *(const float4 *)(s_w + t * WROW + 32 * BK + b_off[q]);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b0.x, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b1.x, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b0.y, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b1.y, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b0.z, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b1.z, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b0.w, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b1.w, acc1, 0, 0, 0);
                } else {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b0.x, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b0.y, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b0.z, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b0.w, acc0, 0, 0, 0);
                }
            }
        }
    };

    // Prefetches are unconditional (the last stage re-fetches its own operands and drops them): a
    // conditionally written register array would be demoted to scratch memory.
    if (TLO == 0 && THI == 3) {
#ifdef MVX_GATHER_STAMPS
    const bool stamped = tile == g_stamp_unit[0] && d == g_stamp_unit[1] && nb == g_stamp_unit[2];
    bool follow = false;
    if (threadIdx.x == 0 && !stamped && g_stamp_wg == (int)blockIdx.x) {
        follow = true;
        g_stamp_wg = -1;
        g_stamps[201] = __builtin_amdgcn_s_memtime();          // the next unit of the same workgroup enters its K loop
    }
#endif
    MVX_STAMP(0);
    load_wrow(0, 0);
    load_halo(0);
    for (int st = 0; st < nstages; ++st) {
        const int nxt = st + 1 < nstages ? st + 1 : st;
        __syncthreads();                           // previous stage's LDS reads are done
        MVX_STAMP(1 + 6 * st);
        store_halo();
        store_wrow();                              // tap row 0
        __syncthreads();
        MVX_STAMP(2 + 6 * st);
#ifdef MVX_GATHER_STAMPS
        if (follow && st == 0) g_stamps[202] = __builtin_amdgcn_s_memtime();       // ... and has its first operands in LDS
#endif
        load_wrow(st, 1);                          // next weight row first ...
        load_halo(nxt);                            // ... then the long-latency halo of the next stage
        compute_row(0);
        __syncthreads();
        MVX_STAMP(3 + 6 * st);
        store_wrow();                              // tap row 1 (waits for its 6 loads only)
        __syncthreads();
        MVX_STAMP(4 + 6 * st);
        load_wrow(st, 2);
        compute_row(1);
        __syncthreads();
        MVX_STAMP(5 + 6 * st);
        store_wrow();                              // tap row 2
        __syncthreads();
        MVX_STAMP(6 + 6 * st);
        load_wrow(nxt, 0);
        compute_row(2);
    }
#ifdef MVX_GATHER_STAMPS
    __syncthreads();
    MVX_STAMP(1 + 6 * nstages);
#endif
    } else {
    // two tap rows [r0, r0 + 1] (2 x 2 window): the same pipeline with one row step less per stage.  With g.s2d the
    // structurally zero (tap, parity) blocks of a stride-2 kernel in space-to-depth form are skipped: forward (mode 0) the
    // parity is that of the stage's input-channel chunk, dgrad (mode 1, flipped window {1,2}^2) that of the unit's
    // output-channel block; window tap of row a / column t: forward (a, t), dgrad (2 - a, 2 - t).
    constexpr int r0 = TLO;
    auto tap_masks = [&](int st, bool &row0, bool &row1, unsigned &colmask) __attribute__((always_inline)) {
        row0 = row1 = true;
        colmask = 7u;
        if (g.s2d <= 0) return;
        int p;
        if (g.mode == 0) {
            int kd, ds, cc;
            stage_kd(st, kd, ds, cc);
            p = (cc * BK) / g.s2d;
        } else {
            p = (nb * BNW) / g.s2d;
        }
        const unsigned m = s2d_tap_mask(p);            // bit (ta * 2 + tb)
        auto wt = [&](int a) { return g.mode == 0 ? a : 2 - a; };        // window tap of kernel row / column index a
        const int ta0 = wt(r0), ta1 = wt(r0 + 1);
        colmask = 0u;
        bool any0 = false, any1 = false;
#pragma unroll
        for (int t = TLO; t < THI; ++t) {
            const int tb = wt(t);
            const bool v0 = (m >> (ta0 * 2 + tb)) & 1u, v1 = (m >> (ta1 * 2 + tb)) & 1u;
            // a column is executed if either row needs it; rows are switched off as a whole below (the valid columns of
            // the two rows coincide wherever both rows are valid: validity factorises into a row and a column condition)
            if (v0 || v1) colmask |= 1u << t;
            any0 |= v0;
            any1 |= v1;
        }
        row0 = any0;
        row1 = any1;
    };
    load_wrow(0, r0);
    load_halo(0);
    for (int st = 0; st < nstages; ++st) {
        const int nxt = st + 1 < nstages ? st + 1 : st;
        bool row0, row1;
        unsigned colmask;
        tap_masks(st, row0, row1, colmask);
        __syncthreads();
        store_halo();
        store_wrow();                              // tap row r0
        __syncthreads();
        load_wrow(st, r0 + 1);
        load_halo(nxt);
        if (row0) compute_row(r0, colmask);
        __syncthreads();
        store_wrow();                              // tap row r0 + 1
        __syncthreads();
        load_wrow(nxt, r0);
        if (row1) compute_row(r0 + 1, colmask);
    }
    }
    }   // active

    // ---- epilogue: bias, ReLU, store, BatchNorm statistics (identical to conv3d_gather)
    const int n0 = nb * BNW + li, n1 = NB2 == 2 ? n0 + 32 : n0;      // narrow unit: the second channel tile does not exist
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
    float skip0 = 0.f, skip1 = 0.f;            // constants of the depth taps that were not executed (see the stage list)
    if (skipped && active) {
        const float *bg_tap = bg_pre + (size_t)g.Dout * g.F * g.Cout + (size_t)d * 3 * g.Cout;
#pragma unroll
        for (int kd = 0; kd < 3; ++kd)
            if ((skipped >> kd) & 1u) { skip0 += bg_tap[kd * g.Cout + n0]; skip1 += bg_tap[kd * g.Cout + n1]; }
    }
    // background value of this plane: the same fp32 operations as a computed site, on the constant
    float bgv0 = (bg_pre ? bg_pre[(size_t)d * g.Cout + n0] : 0.f) + bias0, bgv1 = (bg_pre ? bg_pre[(size_t)d * g.Cout + n1] : 0.f) + bias1;
    if (relu) { bgv0 = fmaxf(bgv0, 0.f); bgv1 = fmaxf(bgv1, 0.f); }
    // BatchNorm sums in f64 from the first addition on: var = E[y^2] - mean^2 cancels, and f32 partial sums cost
    // mean^2 / var times their 1e-7 in the variance (the reference's torch-CPU BatchNorm accumulates in double too)
    // the 16 site-mask bytes of this lane are fetched up front, unconditionally from clamped coordinates: a load between the
    // stores (vector-memory operations return in order) made every store pair wait for the previous one to complete
    unsigned site_on = active ? 0xffffu : 0u;          // bit r: site r is a computed (non-background) site
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
    double s1a = 0.0, s2a = 0.0, s1b = 0.0, s2b = 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int gy = ty0 + 2 * wv + (row >> 4), gx = tx0 + (row & 15);
        float v0 = (acc0[r] + skip0) + bias0, v1 = (acc1[r] + skip1) + bias1;
        if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
        // a background SITE holds the constant bit for bit, also inside a computed tile
        if (out_mask && !((site_on >> r) & 1u)) { v0 = bgv0; v1 = bgv1; }
        if (gy < g.H && gx < g.W) {
            float *o = out + (((size_t)d * g.H + gy) * g.W + gx) * g.Cout;
            o[n0] = v0;
            if (NB2 == 2) o[n1] = v1;
            s1a += (double)v0; s2a += (double)v0 * (double)v0;
            s1b += (double)v1; s2b += (double)v1 * (double)v1;
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
        if (tid < 2 * BNW) {
            const int which = tid / BNW, c = tid % BNW, j = which * BN + c;
            const double t = s_red[0][j] + s_red[1][j] + s_red[2][j] + s_red[3][j];
            const unsigned rep = (unsigned)(tile + d * ntiles) % MVX_REP;
            double *fstats = stats + (size_t)(d / g.Dout) * MVX_REP * 2 * g.Cout;       // the plane's frame
            atomicAdd(fstats + ((size_t)rep * 2 + which) * g.Cout + nb * BNW + c, t);
        }
    }
#ifdef MVX_GATHER_STAMPS
    if (TLO == 0 && THI == 3 && threadIdx.x == 0 && tile == g_stamp_unit[0] && d == g_stamp_unit[1] && nb == g_stamp_unit[2]) {
        __builtin_amdgcn_s_waitcnt(0);                          // the unit's stores have been issued and its loads returned
        g_stamps[200] = __builtin_amdgcn_s_memtime();
        g_stamp_wg = (int)blockIdx.x;
    }
#endif
}



// Classic launch: one unit per workgroup, grid = (tiles, planes, channel blocks).
template <int TLO, int THI, int NB2 = 2>
__global__ __launch_bounds__(256, 3) void conv3d_gather_pf(const float *__restrict__ in,
                                                           const float *__restrict__ wpk,
                                                           const float *__restrict__ bias,
                                                           float *__restrict__ out, double *__restrict__ stats,
                                                           Geom g, int relu, const int *__restrict__ in_hflag,
                                                           const unsigned char *__restrict__ out_mask,
                                                           const float *__restrict__ bg_pre, int border_active,
                                                           unsigned long long *__restrict__ exec_stages,
                                                           const int *__restrict__ only_tiles,
                                                           unsigned *__restrict__ done_counter, double fin_count,
                                                           double fin_eps, float *__restrict__ fin_mean_inv) {
    __shared__ __attribute__((aligned(16))) float s_halo[HH * HROW];
    __shared__ __attribute__((aligned(16))) float s_w[3 * WROW];
    double (*s_red)[2 * BN] = reinterpret_cast<double (*)[2 * BN]>(s_w);      // epilogue scratch: the weights are done by then
    gather_unit<TLO, THI, NB2>(blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, s_halo, s_w, s_red, in, wpk, bias, out, stats, g,
                               relu, in_hflag, out_mask, bg_pre, border_active, exec_stages, only_tiles);
    if (stats && done_counter) {
        __shared__ int s_last;
        bn_finalize_by_last_block(done_counter, gridDim.x * gridDim.y * gridDim.z, stats, g.Cout, fin_count, fin_eps,
                                  fin_mean_inv, &s_last, g.F);
    }
}

// Persistent launch: gridDim.x workgroups (two per CU) pull units from a device counter until none is left -- every
// workgroup reaches the exit (`u >= units`), so the grid always drains.  Units are handed out in the classic order
// (tile fastest), dynamically: no tail round with idle CUs (3,300 units over 512 slots = 6.45 rounds) and background
// tiles, which only write a constant, do not unbalance the workgroups.
template <int TLO, int THI, int NB2 = 2>
__global__ __launch_bounds__(256, 3) void conv3d_gather_pw(const float *__restrict__ in,
                                                           const float *__restrict__ wpk,
                                                           const float *__restrict__ bias,
                                                           float *__restrict__ out, double *__restrict__ stats,
                                                           Geom g, int relu, const int *__restrict__ in_hflag,
                                                           const unsigned char *__restrict__ out_mask,
                                                           const float *__restrict__ bg_pre, int border_active,
                                                           unsigned long long *__restrict__ exec_stages,
                                                           const int *__restrict__ only_tiles,
                                                           unsigned *__restrict__ done_counter, double fin_count,
                                                           double fin_eps, float *__restrict__ fin_mean_inv,
                                                           unsigned *__restrict__ work_counter, int ntiles, int nplanes,
                                                           int nblocks) {
    __shared__ __attribute__((aligned(16))) float s_halo[HH * HROW];
    __shared__ __attribute__((aligned(16))) float s_w[3 * WROW];
    double (*s_red)[2 * BN] = reinterpret_cast<double (*)[2 * BN]>(s_w);      // epilogue scratch: the weights are done by then
    __shared__ unsigned s_unit;
    const unsigned units = (unsigned)ntiles * nplanes * nblocks;
    for (;;) {
        __syncthreads();                            // everyone is done with the previous unit (LDS, s_unit)
        if (threadIdx.x == 0) s_unit = atomicAdd(work_counter, 1u);
        __syncthreads();
        const unsigned u = s_unit;
        if (u >= units) break;                      // block-uniform
        const int tile = u % ntiles, d = (u / ntiles) % nplanes, nb = u / (ntiles * nplanes);
        gather_unit<TLO, THI, NB2>(tile, d, nb, ntiles, s_halo, s_w, s_red, in, wpk, bias, out, stats, g, relu, in_hflag, out_mask,
                                   bg_pre, border_active, exec_stages, only_tiles);
    }
    if (stats && done_counter) {
        __shared__ int s_last;
        bn_finalize_by_last_block(done_counter, gridDim.x, stats, g.Cout, fin_count, fin_eps, fin_mean_inv, &s_last, g.F);
    }
}

// ------------------------------------------------------------------------------------------
// weight gradient
// ------------------------------------------------------------------------------------------
constexpr int WG_THREADS = 9 * 64;
constexpr int XP = BK;                 // halo pitch (conflict-free for lane-consecutive b32 reads)
constexpr int ZP = BN;                 // dz pitch

__global__ __launch_bounds__(WG_THREADS) void conv3d_wgrad(const float *__restrict__ in,
                                                           const float *__restrict__ dz,
                                                           float *__restrict__ slabs, Geom g,
                                                           int tiles_per_strip) {
    __shared__ __attribute__((aligned(16))) float s_x[HH * HW * XP];
    __shared__ __attribute__((aligned(16))) float s_z[TH * TW * ZP];
    const int tiles_x = (g.W + TW - 1) / TW, tiles_y = (g.H + TH - 1) / TH;
    const int ntiles = tiles_x * tiles_y;
    const int strip = blockIdx.x;
    const int nchunks = g.Cin / BK;
    const int kd = blockIdx.y / nchunks, cc = blockIdx.y % nchunks;
    const int tid = threadIdx.x, lane = tid & 63, tap = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int ta = tap / 3, tb = tap % 3;

    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }

    const int t_beg = strip * tiles_per_strip;
    const int t_end = min(ntiles, t_beg + tiles_per_strip);
    // (depth plane, patch) work list of this workgroup, walked with a one-step register prefetch:
    // the next patch's halo and dz tile are loaded under the current patch's 128 MFMAs per wave.
    constexpr int NX = (HH * HW * 8 + WG_THREADS - 1) / WG_THREADS;     // float4 per thread, halo
    constexpr int NZ = (TH * TW * 16 + WG_THREADS - 1) / WG_THREADS;    // float4 per thread, dz
    float4 xr[NX], zr[NZ];
    auto load_step = [&](int d, int t) {
        const int ds = d * g.sd - g.pd + kd;
        const int tx0 = (t % tiles_x) * TW, ty0 = (t / tiles_x) * TH;
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            const int c = tid + WG_THREADS * u;
            xr[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < HH * HW * 8) {
                const int r = c >> 3, part = c & 7;
                const int gy = ty0 - 1 + r / HW, gx = tx0 - 1 + r % HW;
                if (gy >= 0 && gy < g.H && gx >= 0 && gx < g.W)
                    xr[u] = *(const float4 *)(in + (((size_t)ds * g.H + gy) * g.W + gx) * g.Cin + cc * BK + part * 4);
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
    // first valid output plane for this depth tap
    auto next_valid = [&](int d) {
        while (d < g.Dout) {
            const int ds = d * g.sd - g.pd + kd;
            if (ds >= 0 && ds < g.Din) break;
            ++d;
        }
        return d;
    };
    int d = next_valid(0), t = t_beg;
    const bool any = d < g.Dout && t_beg < t_end;
    if (any) load_step(d, t);
    while (any && d < g.Dout) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            const int c = tid + WG_THREADS * u;
            if (c < HH * HW * 8) *(float4 *)(s_x + (c >> 3) * XP + (c & 7) * 4) = xr[u];
        }
#pragma unroll
        for (int u = 0; u < NZ; ++u) {
            const int c = tid + WG_THREADS * u;
            if (c < TH * TW * 16) *(float4 *)(s_z + (c >> 4) * ZP + (c & 15) * 4) = zr[u];
        }
        __syncthreads();
        // advance the work list and prefetch
        int nt = t + 1, nd = d;
        if (nt >= t_end) { nt = t_beg; nd = next_valid(d + 1); }
        if (nd < g.Dout) load_step(nd, nt);
#pragma unroll 4
        for (int kk = 0; kk < TH * TW / 2; ++kk) {
            const int s = 2 * kk + lh;
            const float a = s_x[(((s >> 4) + ta) * HW + (s & 15) + tb) * XP + li];
            const float b0 = s_z[s * ZP + li], b1 = s_z[s * ZP + 32 + li];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
        }
        t = nt; d = nd;
    }
    // slab[strip][kd][tap][c (Cin)][n (64)]
    float *o = slabs + ((((size_t)strip * 3 + kd) * 9 + tap) * g.Cin + cc * BK) * BN;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        o[(size_t)row * BN + li] = acc0[r];
        o[(size_t)row * BN + 32 + li] = acc1[r];
    }
}


// ------------------------------------------------------------------------------------------
// weight gradient, SIMD-balanced form (used when Cin % 64 == 0).  A workgroup has EIGHT waves: (m, n) in 2 x 2 over a
// 64(c) x 64(n) block of dW, times two tap groups -- waves 0-3 walk in-plane taps {0..4}, waves 4-7 taps {5..8} of the
// same staged tile (2-tap / 2-tap for the 2x2 window of the stride-2 RPN layers).  Waves w and w+4 share a SIMD, so every
// SIMD carries nine MFMAs per k step and two waves to hide each other's LDS and barrier waits: the earlier four-wave form
// (all nine taps per wave, 144 accumulator + 80 prefetch VGPRs = 300, ONE wave per SIMD) ran at 0.46 of the matrix peak
// in isolation.  Here: 80 accumulator + 40 prefetch VGPRs, one workgroup (2 waves per SIMD) per CU.
// ------------------------------------------------------------------------------------------
struct Strips { int n[3]; };       // workgroups (strips) per depth tap in list mode
constexpr int W4_THREADS = 512;
constexpr int W4_C = 64;               // input channels per workgroup

// tap OWNED (written) by slot i of a tap group (-1: none); the first w4_ncomp() slots are computed, the rest stay zero
__host__ __device__ constexpr int w4_own(bool t2, int grp, int i) {
    return t2 ? (grp == 0 ? (i < 3 ? i : i + 2) : (i < 2 ? i + 3 : (i < 4 ? i + 5 : -1)))      // {0,1,2,5,6} / {3,4,7,8}
              : (grp == 0 ? i : (i < 4 ? i + 5 : -1));                                         // {0,1,2,3,4} / {5,6,7,8}
}
__host__ __device__ constexpr int w4_ncomp(bool t2, int grp) { return t2 ? 2 : (grp == 0 ? 5 : 4); }

// ``slots``: bit i = slot i of the group is executed (block-uniform; all ones except for the structurally zero (tap, parity)
// blocks of a stride-2 kernel in space-to-depth form, Geom::s2d)
template <bool T2, int GRP>
__device__ __forceinline__ void wgrad4_mfma_step(const float *__restrict__ s_x, const float *__restrict__ s_z, f32x16 (&acc)[5],
                                                 int wm, int wn, int li, int lh, unsigned slots) {
    if (T2 && slots == 0u) return;
#pragma unroll 2
    for (int kk = 0; kk < TH * TW / 2; ++kk) {
        const int s = 2 * kk + lh;
        const float b = s_z[s * ZP + wn * 32 + li];
        const float *xa = s_x + ((s >> 4) * HW + (s & 15)) * W4_C + wm * 32 + li;
#pragma unroll
        for (int i = 0; i < w4_ncomp(T2, GRP); ++i) {
            if (T2 && !((slots >> i) & 1u)) continue;
            const int t9 = w4_own(T2, GRP, i);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[((t9 / 3) * HW + (t9 % 3)) * W4_C], b, acc[i], 0, 0, 0);
        }
    }
}

template <bool T2>
__global__ __launch_bounds__(W4_THREADS) void conv3d_wgrad4(const float *__restrict__ in,
                                                            const float *__restrict__ dz,
                                                            float *__restrict__ slabs, Geom g,
                                                            int tiles_per_strip, const int *__restrict__ step_list,
                                                            const int *__restrict__ step_count,
                                                            const float *__restrict__ c_in, Strips ks) {
    __shared__ __attribute__((aligned(16))) float s_x[HH * HW * W4_C];
    __shared__ __attribute__((aligned(16))) float s_z[TH * TW * ZP];
    const int tiles_x = (g.W + TW - 1) / TW, tiles_y = (g.H + TH - 1) / TH;
    const int ntiles = tiles_x * tiles_y;
    const int nchunks = g.Cin / W4_C;
    const int kd = blockIdx.y / nchunks, cc = blockIdx.y % nchunks;
    // strips of this depth tap: the list mode hands each tap a share of the workgroups proportional to its number of
    // valid planes (conv3: 1 / 2 / 1); the dense mode uses the whole grid for every tap
    const int strip = blockIdx.x, nstrips = step_list ? (kd == 0 ? ks.n[0] : (kd == 1 ? ks.n[1] : ks.n[2])) : (int)gridDim.x;
    if (strip >= nstrips) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int grp = wv >> 2;                         // tap group
    const int wm = (wv >> 1) & 1, wn = wv & 1;
    const int nb = blockIdx.z;                       // 64-channel block of dz / dW (Cout = 64 * gridDim.z)
    dz += (size_t)nb * BN;
    slabs += (size_t)nb * gridDim.x * 27 * g.Cin * BN;       // slab index = blockIdx.x (gridDim.x = the largest share)

    f32x16 acc[5];
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // Steps of this workgroup: (output plane d with a valid source plane for kd) x (tile).
    // Dense: the strip's contiguous tile range for every valid plane.  With a background description the
    // steps whose source halo holds a non-background site were compacted into step_list[kd][] (everywhere
    // else x - c is exactly zero) and are dealt round-robin to the strips: balanced and deterministic.
    int dlo = 0, dhi = -1;                           // valid output planes form a contiguous range
    for (int d = 0; d < g.Dout; ++d) {
        const int ds = d * g.sd - g.pd + kd;
        if (ds >= 0 && ds < g.Din) { if (dhi < 0) dlo = d; dhi = d; }
    }
    const int nd = dhi >= dlo ? dhi - dlo + 1 : 0;
    const int per = tiles_per_strip;
    const int *my_list = step_list ? step_list + (size_t)kd * g.Dout * g.F * ntiles : nullptr;
    const int nlist = step_list ? step_count[kd] : 0;
    const int nsteps = step_list ? (nlist > strip ? (nlist - strip + nstrips - 1) / nstrips : 0) : nd * per;
    // step i -> (plane, tile); dense steps past the last tile are dead (ragged last strip)
    auto step_of = [&](int i, int &d, int &t) {
        if (my_list) { const int e = my_list[strip + i * nstrips]; d = e / ntiles; t = e - d * ntiles; }
        else { d = dlo + i / per; t = strip * per + i % per; }
    };

    constexpr int NX = (HH * HW * 16 + W4_THREADS - 1) / W4_THREADS;    // float4 per thread, halo (64 ch)
    constexpr int NZ = TH * TW * 16 / W4_THREADS;                       // float4 per thread, dz
    f32x4 xr[NX], zr[NZ];
    auto load_step = [&](int i) __attribute__((always_inline)) {
        int d, t;
        step_of(i, d, t);
        const int ds = mvx_src_plane(d, g.Din, g.Dout, g.sd, g.pd, kd);      // listed / dense steps always have a valid source
        const int tx0 = (t % tiles_x) * TW, ty0 = (t / tiles_x) * TH;
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            const int c = tid + W4_THREADS * u;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (c < HH * HW * 16) {
                const int r = c >> 4, part = c & 15;
                const int gy = ty0 - 1 + r / HW, gx = tx0 - 1 + r % HW;
                if (gy >= 0 && gy < g.H && gx >= 0 && gx < g.W) {
                    v = *(const f32x4 *)(in + (((size_t)ds * g.H + gy) * g.W + gx) * g.Cin + cc * W4_C + part * 4);
                    if (c_in) v -= *(const f32x4 *)(c_in + (size_t)ds * g.Cin + cc * W4_C + part * 4);
                }
            }
            xr[u] = v;
        }
#pragma unroll
        for (int u = 0; u < NZ; ++u) {
            const int c = tid + W4_THREADS * u;
            const int r = c >> 4, part = c & 15;
            const int gy = ty0 + (r >> 4), gx = tx0 + (r & 15);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (gy < g.H && gx < g.W)
                v = *(const f32x4 *)(dz + (((size_t)d * g.H + gy) * g.W + gx) * g.Cout + part * 4);
            zr[u] = v;
        }
    };
    auto next_live = [&](int i) {
        if (my_list) return i < nsteps ? i : nsteps;
        while (i < nsteps && strip * per + i % per >= ntiles) ++i;      // ragged last strip
        return i < nsteps ? i : nsteps;
    };
    // stride-2 kernel in space-to-depth form: the window taps that carry weight for this workgroup's parity block
    // (group 0 owns window taps (0,0), (0,1) = slots 0, 1; group 1 owns (1,0), (1,1))
    unsigned slots = 0x1fu;
    if (T2 && g.s2d > 0) {
        const unsigned m = s2d_tap_mask((cc * W4_C) / g.s2d);
        slots = grp == 0 ? (m & 3u) : ((m >> 2) & 3u);
    }
    int cur = next_live(0);
#ifdef MVX_GATHER_STAMPS
    const bool stamped = strip == g_stamp_unit[0] && (int)blockIdx.y == g_stamp_unit[1] && nb == g_stamp_unit[2];
    int sidx = 0;
#endif
    MVX_STAMP(0);
    if (cur < nsteps) load_step(cur);
    while (cur < nsteps) {
        __syncthreads();
#ifdef MVX_GATHER_STAMPS
        MVX_STAMP(1 + 3 * sidx);
#endif
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            const int c = tid + W4_THREADS * u;
            if (c < HH * HW * 16) *(f32x4 *)(s_x + (c >> 4) * W4_C + (c & 15) * 4) = xr[u];
        }
#pragma unroll
        for (int u = 0; u < NZ; ++u) {
            const int c = tid + W4_THREADS * u;
            *(f32x4 *)(s_z + (c >> 4) * ZP + (c & 15) * 4) = zr[u];
        }
        __syncthreads();
#ifdef MVX_GATHER_STAMPS
        MVX_STAMP(2 + 3 * sidx);
#endif
        const int nxt = next_live(cur + 1);
        load_step(nxt < nsteps ? nxt : cur);          // unconditional (see conv3d_gather_pf): the last one is dropped
        if (grp == 0) wgrad4_mfma_step<T2, 0>(s_x, s_z, acc, wm, wn, li, lh, slots);
        else wgrad4_mfma_step<T2, 1>(s_x, s_z, acc, wm, wn, li, lh, slots);
#ifdef MVX_GATHER_STAMPS
        MVX_STAMP(3 + 3 * sidx);                      // thread 0 (wave 0) has ISSUED its MFMAs of the step
        ++sidx;
#endif
        cur = nxt;
    }
#ifdef MVX_GATHER_STAMPS
    __syncthreads();
    MVX_STAMP(250);
    if (stamped && threadIdx.x == 0) g_stamps[251] = (unsigned long long)nsteps;
#endif
    // slab[strip][kd][tap][c (Cin)][n (64)]: every tap is written by the group that owns it (zeros where nothing was computed)
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int t9 = grp == 0 ? w4_own(T2, 0, i) : w4_own(T2, 1, i);
        if (t9 < 0) continue;
        float *o = slabs + ((((size_t)strip * 3 + kd) * 9 + t9) * g.Cin + cc * W4_C + wm * 32) * BN + wn * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            o[(size_t)row * BN + li] = acc[i][r];
        }
    }
#ifdef MVX_GATHER_STAMPS
    if (stamped && threadIdx.x == 0) {
        __builtin_amdgcn_s_waitcnt(0);
        g_stamps[252] = __builtin_amdgcn_s_memtime();
    }
#endif
}

// ------------------------------------------------------------------------------------------
// conv3d_wgrad4s<T2, NP>: conv3d_wgrad4 in split arithmetic on the bf16 matrix cores (split_common.h: NP = 2 "bf16x3",
// NP = 3 "bf16x6" = fp32-grade).  Same workgroup decomposition, step lists, strips and slab layout as conv3d_wgrad4 --
// eight waves, wave (grp, wm, wn) owns the 32(c) x 32(n) block (wm, wn) of the taps of its group, waves w and w + 4 share
// a SIMD so that every SIMD carries nine taps per k step -- so the host code and wgrad_reduce are shared.  The MFMA
// reduction index is the SITE while memory is channel-major: the 10 x 18-site halo of x and the 8 x 16-site tile of dz are
// cut into their bf16 pieces while they are staged as [piece][32-channel block][site][32] (64-byte rows) and fetched with
// ds_read_b64_tr_b16, the LDS transpose read (a 16-lane group reads 4 sites x 16 channels, every lane gets 4 consecutive
// sites of its channel = half of a 32x32x16 operand fragment); a tap shift only changes the rows addressed.  Per 16-site k
// step a wave reads NP fragments of dz once and NP fragments of x per tap: (taps + 1) NP fragments for taps x 3 (6) MFMAs.
// LDS 79 KB (NP = 2) / 118 KB (NP = 3): one workgroup, two waves per SIMD, per CU.
// ------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8 w4s_frag(const unsigned short *row0, const unsigned short *row1) {
    typedef __attribute__((address_space(3))) s16x4 lds4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4 *)row0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4 *)row1);
    s16x8 v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
    v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(bf16x8, v);
}

// One staged tile (8 x 16 sites = eight 16-site k steps) of conv3d_wgrad4s for tap group GRP (compile time, like
// wgrad4_mfma_step: the tap offsets are constants and the loop unrolls without branches).  All operand fragments of a k step
// are fetched before its MFMAs, so the transpose reads of one tap are in flight under the MFMAs of the previous one.
template <bool T2, int NP, int GRP, int FMT>
__device__ __forceinline__ void wgrad4s_mfma_step(const unsigned short (*__restrict__ s_x)[2][HH * HW][32],
                                                  const unsigned short (*__restrict__ s_z)[2][TH * TW][32], f32x16 (&acc)[5],
                                                  int wm, int wn, int pcol, int kq, unsigned slots) {
    if (T2 && slots == 0u) return;
    constexpr int NC = w4_ncomp(T2, GRP);
#pragma unroll 2
    for (int ks16 = 0; ks16 < TH; ++ks16) {       // 16 sites (one patch row) per MFMA k step
        const int zr0 = ks16 * TW + kq, zr1 = zr0 + 4;
        bf16x8 bz[NP], ax[NC][NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) bz[p] = w4s_frag(&s_z[p][wn][zr0][pcol], &s_z[p][wn][zr1][pcol]);
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int t9 = w4_own(T2, GRP, i);
            const int hr0 = (ks16 + t9 / 3) * HW + (t9 % 3) + kq, hr1 = hr0 + 4;
#pragma unroll
            for (int p = 0; p < NP; ++p) ax[i][p] = w4s_frag(&s_x[p][wm][hr0][pcol], &s_x[p][wm][hr1][pcol]);
        }
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            if (T2 && !((slots >> i) & 1u)) continue;
            split_mac1<NP, FMT>(acc[i], ax[i], bz);
        }
    }
}

template <bool T2, int NP, int FMT>
__global__ __launch_bounds__(W4_THREADS) void conv3d_wgrad4s(const float *__restrict__ in,
                                                             const float *__restrict__ dz,
                                                             float *__restrict__ slabs, Geom g,
                                                             int tiles_per_strip, const int *__restrict__ step_list,
                                                             const int *__restrict__ step_count,
                                                             const float *__restrict__ c_in, Strips ks, SplitAmax am) {
    float x_scale = 1.f, z_scale = 1.f;                       // fp16 pieces: operands scaled by their bound amax (split_common.h)
    if constexpr (FMT == 1) { x_scale = split_scale_coarse(am.a); z_scale = split_scale_of(am.b); }     // x: activations, dz: gradients
    __shared__ __attribute__((aligned(16))) unsigned short s_x[NP][2][HH * HW][32];      // [piece][32-channel block][halo site][channel]
    __shared__ __attribute__((aligned(16))) unsigned short s_z[NP][2][TH * TW][32];      // [piece][32-channel block][site][channel]
    const int tiles_x = (g.W + TW - 1) / TW, tiles_y = (g.H + TH - 1) / TH;
    const int ntiles = tiles_x * tiles_y;
    const int nchunks = g.Cin / W4_C;
    const int kd = blockIdx.y / nchunks, cc = blockIdx.y % nchunks;
    const int strip = blockIdx.x, nstrips = step_list ? (kd == 0 ? ks.n[0] : (kd == 1 ? ks.n[1] : ks.n[2])) : (int)gridDim.x;
    if (strip >= nstrips) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int grp = wv >> 2;                         // tap group
    const int wm = (wv >> 1) & 1, wn = wv & 1;
    const int nb = blockIdx.z;                       // 64-channel block of dz / dW (Cout = 64 * gridDim.z)
    dz += (size_t)nb * BN;
    slabs += (size_t)nb * gridDim.x * 27 * g.Cin * BN;
    // transpose-read roles of this lane: 16-lane group -> (k half, 16-column half), lane -> (4-site row group, 4 columns)
    const int g16 = lane >> 4, i16 = lane & 15, q = i16 >> 2, pcol = (g16 & 1) * 16 + 4 * (i16 & 3), kbase = (g16 >> 1) * 8;

    f32x16 acc[5];
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    int dlo = 0, dhi = -1;                           // valid output planes form a contiguous range
    for (int d = 0; d < g.Dout; ++d) {
        const int ds = d * g.sd - g.pd + kd;
        if (ds >= 0 && ds < g.Din) { if (dhi < 0) dlo = d; dhi = d; }
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
    constexpr int NX = (HH * HW * 16 + W4_THREADS - 1) / W4_THREADS;    // float4 per thread, halo (64 ch)
    constexpr int NZ = TH * TW * 16 / W4_THREADS;                       // float4 per thread, dz
    f32x4 xr[NX], zr[NZ];
    auto load_step = [&](int i) __attribute__((always_inline)) {
        int d, t;
        step_of(i, d, t);
        const int ds = mvx_src_plane(d, g.Din, g.Dout, g.sd, g.pd, kd);      // listed / dense steps always have a valid source
        const int tx0 = (t % tiles_x) * TW, ty0 = (t / tiles_x) * TH;
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            const int c = tid + W4_THREADS * u;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (c < HH * HW * 16) {
                const int r = c >> 4, part = c & 15;
                const int gy = ty0 - 1 + r / HW, gx = tx0 - 1 + r % HW;
                if (gy >= 0 && gy < g.H && gx >= 0 && gx < g.W) {
                    v = *(const f32x4 *)(in + (((size_t)ds * g.H + gy) * g.W + gx) * g.Cin + cc * W4_C + part * 4);
                    if (c_in) v -= *(const f32x4 *)(c_in + (size_t)ds * g.Cin + cc * W4_C + part * 4);
                }
            }
            xr[u] = v;
        }
#pragma unroll
        for (int u = 0; u < NZ; ++u) {
            const int c = tid + W4_THREADS * u;
            const int r = c >> 4, part = c & 15;
            const int gy = ty0 + (r >> 4), gx = tx0 + (r & 15);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (gy < g.H && gx < g.W)
                v = *(const f32x4 *)(dz + (((size_t)d * g.H + gy) * g.W + gx) * g.Cout + part * 4);
            zr[u] = v;
        }
    };
    auto next_live = [&](int i) {
        if (my_list) return i < nsteps ? i : nsteps;
        while (i < nsteps && strip * per + i % per >= ntiles) ++i;      // ragged last strip
        return i < nsteps ? i : nsteps;
    };
    unsigned slots = 0x1fu;                          // see conv3d_wgrad4: the window taps that carry weight for this parity block
    if (T2 && g.s2d > 0) {
        const unsigned m = s2d_tap_mask((cc * W4_C) / g.s2d);
        slots = grp == 0 ? (m & 3u) : ((m >> 2) & 3u);
    }
    int cur = next_live(0);
    if (cur < nsteps) load_step(cur);
    while (cur < nsteps) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            const int c = tid + W4_THREADS * u;
            if (c < HH * HW * 16) {
                const int r = c >> 4, part = c & 15;
                uint2 pc[NP];
                if constexpr (FMT == 1) xr[u] *= x_scale;
                split_n<NP, FMT>(xr[u][0], xr[u][1], xr[u][2], xr[u][3], pc);
#pragma unroll
                for (int p = 0; p < NP; ++p) *(uint2 *)(&s_x[p][part >> 3][r][(part & 7) * 4]) = pc[p];
            }
        }
#pragma unroll
        for (int u = 0; u < NZ; ++u) {
            const int c = tid + W4_THREADS * u;
            const int r = c >> 4, part = c & 15;
            uint2 pc[NP];
            if constexpr (FMT == 1) zr[u] *= z_scale;
            split_n<NP, FMT>(zr[u][0], zr[u][1], zr[u][2], zr[u][3], pc);
#pragma unroll
            for (int p = 0; p < NP; ++p) *(uint2 *)(&s_z[p][part >> 3][r][(part & 7) * 4]) = pc[p];
        }
        __syncthreads();
        const int nxt = next_live(cur + 1);
        load_step(nxt < nsteps ? nxt : cur);          // unconditional (see conv3d_gather_pf): the last one is dropped
        if (grp == 0) wgrad4s_mfma_step<T2, NP, 0, FMT>(s_x, s_z, acc, wm, wn, pcol, kbase + q, slots);
        else wgrad4s_mfma_step<T2, NP, 1, FMT>(s_x, s_z, acc, wm, wn, pcol, kbase + q, slots);
        cur = nxt;
    }
    float o_scale = 1.f;
    if constexpr (FMT == 1) o_scale = split_inverse(x_scale) * split_inverse(z_scale);
    // slab[strip][kd][tap][c (Cin)][n (64)]: every tap is written by the group that owns it (zeros where nothing was computed)
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int t9 = grp == 0 ? w4_own(T2, 0, i) : w4_own(T2, 1, i);
        if (t9 < 0) continue;
        float *o = slabs + ((((size_t)strip * 3 + kd) * 9 + t9) * g.Cin + cc * W4_C + wm * 32) * BN + wn * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            o[(size_t)row * BN + li] = FMT == 1 ? acc[i][r] * o_scale : acc[i][r];
        }
    }
}

// conv3d_wgrad4 in the arithmetic the flags ask for: exact f32, or the split forms (MVX_FLAG_SPLIT: bf16x3, + MVX_FLAG_SPLIT3: bf16x6)
template <bool T2>
static void launch_wgrad4(int flags, dim3 grid, hipStream_t st, const float *in, const float *dz, float *slabs, const Geom &g,
                          int per, const int *list, const int *count, const float *c_in, const Strips &ks, const SplitAmax &am) {
    if ((flags & MVX_FLAG_SPLIT) && (flags & MVX_FLAG_SPLIT_F16))
        hipLaunchKernelGGL((conv3d_wgrad4s<T2, 2, 1>), grid, dim3(W4_THREADS), 0, st, in, dz, slabs, g, per, list, count, c_in, ks, am);
    else if (flags & MVX_FLAG_SPLIT3)
        hipLaunchKernelGGL((conv3d_wgrad4s<T2, 3, 0>), grid, dim3(W4_THREADS), 0, st, in, dz, slabs, g, per, list, count, c_in, ks, am);
    else if (flags & MVX_FLAG_SPLIT)
        hipLaunchKernelGGL((conv3d_wgrad4s<T2, 2, 0>), grid, dim3(W4_THREADS), 0, st, in, dz, slabs, g, per, list, count, c_in, ks, am);
    else
        hipLaunchKernelGGL(conv3d_wgrad4<T2>, grid, dim3(W4_THREADS), 0, st, in, dz, slabs, g, per, list, count, c_in, ks);
}

// step_list[kd][j] = d * ntiles + tile for the (plane, tile) steps of depth tap kd whose source halo holds a
// non-background site, in ascending (d, tile) order; step_count[kd] = how many.  One workgroup per kd.
__global__ __launch_bounds__(1024) void wgrad_step_list(const int *__restrict__ in_hflag, Geom g, int ntiles,
                                                        int *__restrict__ step_list, int *__restrict__ step_count) {
    __shared__ int smem[17];
    const int total = g.Dout * g.F * ntiles;
    const int kd = blockIdx.x;                     // one workgroup per depth tap
    // every thread takes a contiguous run of entries (its flags in one 64-bit word), ONE block scan per 65,536 entries:
    // the chunk-of-1024 form paid a global load latency and three barriers per chunk (22 chunks: 120 us on the hot path)
    int base = 0;
    for (int s0 = 0; s0 < total; s0 += 1024 * 64) {
        const int left = total - s0;
        const int per = left >= 1024 * 64 ? 64 : (left + 1023) / 1024;
        const int b0 = s0 + (int)threadIdx.x * per;
        unsigned long long bits = 0ull;
#pragma unroll 8
        for (int k = 0; k < per; ++k) {
            const int e = b0 + k;
            if (e < total) {
                const int d = e / ntiles, t = e - d * ntiles;
                const int ds = mvx_src_plane(d, g.Din, g.Dout, g.sd, g.pd, kd);
                const int on = ds >= 0 ? (in_hflag[(size_t)ds * ntiles + t] != 0) : 0;
                bits |= (unsigned long long)on << k;
            }
        }
        int tot;
        int pos = base + block_excl_scan_i32(__popcll(bits), smem, &tot);
        int *dst = step_list + (size_t)kd * total;
        while (bits) {
            const int k = __ffsll((long long)bits) - 1;
            bits &= bits - 1;
            dst[pos++] = b0 + k;
        }
        base += tot;
    }
    if (threadIdx.x == 0) step_count[kd] = base;
}

// ---- closed-form part of the background rewrite of wgrad ------------------------------------------
// R[rep][d][kind][n]: per-plane sums of dz over  0 all sites, 1 row 0, 2 row H-1, 3 column 0, 4 column W-1,
// 5..8 the corners (0,0) (0,W-1) (H-1,0) (H-1,W-1).  One workgroup per (row, plane).
constexpr int RK = 9;
constexpr int RREP = 8;     // replicas of the region sums (spreads the same-address f64 atomics of 1,200 workgroups)
__global__ __launch_bounds__(256) void plane_region_sums(const float *__restrict__ dz, int D, int H, int W, int C,
                                                         double *__restrict__ R) {
    __shared__ float red[256][4];
    const int y = blockIdx.x, d = blockIdx.y;
    const int c4n = C >> 2, ct = threadIdx.x % c4n, st = threadIdx.x / c4n, spb = 256 / c4n;
    const float *row = dz + ((size_t)d * H + y) * W * C;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int x = st; x < W; x += spb) {
        const float4 v = *(const float4 *)(row + (size_t)x * C + ct * 4);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    red[threadIdx.x][0] = s.x; red[threadIdx.x][1] = s.y; red[threadIdx.x][2] = s.z; red[threadIdx.x][3] = s.w;
    __syncthreads();
    if (st == 0) {
        const unsigned rep = (unsigned)y % RREP;
        double *Rp = R + ((size_t)rep * D + d) * RK * C;
        const float4 first = *(const float4 *)(row + ct * 4), last = *(const float4 *)(row + (size_t)(W - 1) * C + ct * 4);
        const float f[4] = {first.x, first.y, first.z, first.w}, l[4] = {last.x, last.y, last.z, last.w};
        for (int j = 0; j < 4; ++j) {
            double t = 0.0;
            for (int q = 0; q < spb; ++q) t += (double)red[q * c4n + ct][j];
            const int n = ct * 4 + j;
            atomicAdd(Rp + 0 * C + n, t);
            if (y == 0) atomicAdd(Rp + 1 * C + n, t);
            if (y == H - 1) atomicAdd(Rp + 2 * C + n, t);
            atomicAdd(Rp + 3 * C + n, (double)f[j]);
            atomicAdd(Rp + 4 * C + n, (double)l[j]);
            if (y == 0) { atomicAdd(Rp + 5 * C + n, (double)f[j]); atomicAdd(Rp + 6 * C + n, (double)l[j]); }
            if (y == H - 1) { atomicAdd(Rp + 7 * C + n, (double)f[j]); atomicAdd(Rp + 8 * C + n, (double)l[j]); }
        }
    }
}

// The same sums over the flagged tiles only (dz is not defined elsewhere): one workgroup per tile, thread = (site
// column group, channel quad); interior tiles only feed kind 0.
__global__ __launch_bounds__(256) void plane_region_sums_tiles(const float *__restrict__ dz, const int *__restrict__ tile_flags,
                                                               int D, int H, int W, int C, double *__restrict__ R) {
    __shared__ float red[256][4];
    const int tiles_x = (W + TW - 1) / TW;
    const int t = blockIdx.x, d = blockIdx.y;
    if (!tile_flags[(size_t)d * gridDim.x + t]) return;
    const int ty0 = (t / tiles_x) * TH, tx0 = (t % tiles_x) * TW;
    const int c4n = C >> 2, ct = threadIdx.x % c4n, st = threadIdx.x / c4n, spb = 256 / c4n;
    const unsigned rep = (unsigned)t % RREP;
    double *Rp = R + ((size_t)rep * D + d) * RK * C;
    // kind k of a site (y, x): 0 always; 1 y==0; 2 y==H-1; 3 x==0; 4 x==W-1; 5..8 corners
    for (int kind = 0; kind < RK; ++kind) {
        const bool need = kind == 0 || (kind == 1 && ty0 == 0) || (kind == 2 && ty0 + TH >= H) || (kind == 3 && tx0 == 0) ||
                          (kind == 4 && tx0 + TW >= W) || (kind == 5 && ty0 == 0 && tx0 == 0) ||
                          (kind == 6 && ty0 == 0 && tx0 + TW >= W) || (kind == 7 && ty0 + TH >= H && tx0 == 0) ||
                          (kind == 8 && ty0 + TH >= H && tx0 + TW >= W);
        if (!need) continue;                          // block-uniform
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int sidx = st; sidx < TH * TW; sidx += spb) {
            const int gy = ty0 + sidx / TW, gx = tx0 + sidx % TW;
            if (gy >= H || gx >= W) continue;
            const bool top = gy == 0, bot = gy == H - 1, lef = gx == 0, rig = gx == W - 1;
            const bool in = kind == 0 || (kind == 1 && top) || (kind == 2 && bot) || (kind == 3 && lef) || (kind == 4 && rig) ||
                            (kind == 5 && top && lef) || (kind == 6 && top && rig) || (kind == 7 && bot && lef) ||
                            (kind == 8 && bot && rig);
            if (!in) continue;
            const float4 v = *(const float4 *)(dz + (((size_t)d * H + gy) * W + gx) * C + ct * 4);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        __syncthreads();
        red[threadIdx.x][0] = s.x; red[threadIdx.x][1] = s.y; red[threadIdx.x][2] = s.z; red[threadIdx.x][3] = s.w;
        __syncthreads();
        if (st == 0)
            for (int j = 0; j < 4; ++j) {
                double tt = 0.0;
                for (int q = 0; q < spb; ++q) tt += (double)red[q * c4n + ct][j];
                atomicAdd(Rp + (size_t)kind * C + ct * 4 + j, tt);
            }
    }
}

// T[d][a][b][n] = sum of dz[d][y][x][n] over the sites whose tap (a,b) source (y+a-1, x+b-1) lies inside the image
__global__ void region_tap_sums(const double *__restrict__ R, const float *__restrict__ extra_total, int D, int C,
                                float *__restrict__ T) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= D * 9 * C) return;
    const int n = e % C, tap = (e / C) % 9, d = e / (9 * C);
    const int a = tap / 3, b = tap % 3;
    double k[RK];
    for (int q = 0; q < RK; ++q) {
        double t = 0.0;
        for (int rep = 0; rep < RREP; ++rep) t += R[(((size_t)rep * D + d) * RK + q) * C + n];
        k[q] = t;
    }
    if (extra_total) k[0] += (double)extra_total[(size_t)d * C + n];      // closed-form share of the unvisited tiles
    double t = k[0];
    if (a == 0) t -= k[1];
    if (a == 2) t -= k[2];
    if (b == 0) t -= k[3];
    if (b == 2) t -= k[4];
    if (a == 0 && b == 0) t += k[5];
    if (a == 0 && b == 2) t += k[6];
    if (a == 2 && b == 0) t += k[7];
    if (a == 2 && b == 2) t += k[8];
    T[e] = (float)t;
}

// dW[co][ci][kd][a][b] += sum over output planes d with a valid source plane of c_in[src(d,kd)][ci] * T[d][a][b][co]
__global__ void wgrad_rank1(const float *__restrict__ T, const float *__restrict__ c_in, float *__restrict__ dw, Geom g) {
    const size_t total = (size_t)27 * g.Cin * g.Cout;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int tap = (int)(e % 9), kd = (int)((e / 9) % 3);
        const int ci = (int)((e / 27) % g.Cin), co = (int)(e / ((size_t)27 * g.Cin));
        float s = 0.f;
        for (int d = 0; d < g.Dout * g.F; ++d) {          // every frame: the constants c_in differ per frame
            const int ds = mvx_src_plane(d, g.Din, g.Dout, g.sd, g.pd, kd);
            if (ds < 0) continue;
            s += c_in[(size_t)ds * g.Cin + ci] * T[((size_t)d * 9 + tap) * g.Cout + co];
        }
        dw[e] += s;
    }
}

// dW[co][ci][kd][kh][kw] = sum_strips slab[strip][kd][tap][ci][co]
__global__ void wgrad_reduce(const float *__restrict__ slabs, float *__restrict__ dw, int nstrips, int Ci, int accumulate,
                             int dst2d, Strips ks) {
    const size_t per = (size_t)27 * Ci * BN;
    slabs += (size_t)blockIdx.y * nstrips * per;     // 64-channel block of the output channels
    dw += (size_t)blockIdx.y * (dst2d ? per / 3 : per);
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < per; e += (size_t)gridDim.x * blockDim.x) {
        const int co = (int)(e % BN);
        size_t r = e / BN;
        const int ci = (int)(r % Ci); r /= Ci;
        const int tap = (int)(r % 9);
        const int kd = (int)(r / 9);
        const int nkd = kd == 0 ? ks.n[0] : (kd == 1 ? ks.n[1] : ks.n[2]);
        const int nk = nkd < nstrips ? nkd : nstrips;                // slabs that exist for this depth tap
        float s = 0.f;
        for (int k = 0; k < nk; ++k) s += slabs[(size_t)k * per + e];
        if (dst2d && kd != 1) continue;             // 2-D kernel gradient: the middle depth slice only
        float *dst = dst2d ? dw + (((size_t)co * Ci + ci) * 3 + tap / 3) * 3 + tap % 3
                           : dw + ((((size_t)co * Ci + ci) * 3 + kd) * 3 + tap / 3) * 3 + tap % 3;
        *dst = accumulate ? *dst + s : s;
    }
}

// A[d][c] = sum over the sites of INPUT plane d of the input gradient dx[.][c], from the tap sums T of dz:
// A[d][c] = sum over kd with an output plane d' reading plane d through it, a, b, n of W[n][c][kd][a][b] T[d'][a][b][n]
__global__ __launch_bounds__(64) void input_grad_sums(const float *__restrict__ w, const float *__restrict__ T, Geom g,
                                                      float *__restrict__ A) {
    const int c = blockIdx.x, d = blockIdx.y;        // one wave per (input channel, GLOBAL input plane)
    double s = 0.0;
    for (int kd = 0; kd < 3; ++kd) {
        const int dp = mvx_dst_plane(d, g.Din, g.Dout, g.sd, g.pd, kd);
        if (dp < 0) continue;
        for (int e = threadIdx.x; e < g.Cout * 9; e += 64) {
            const int n = e / 9, k = e - n * 9;
            s += (double)w[(((size_t)n * g.Cin + c) * 3 + kd) * 9 + k] * (double)T[((size_t)dp * 9 + k) * g.Cout + n];
        }
    }
    s = wave_sum_f64(s);
    if (threadIdx.x == 0) A[(size_t)d * g.Cin + c] = (float)s;
}

}  // namespace

extern "C" size_t mvx_conv3d_packed_weight_bytes(int32_t cout, int32_t cin) {
    return (size_t)27 * cout * cin * sizeof(float);
}

extern "C" int mvx_conv3d_pack_weights(const float *w, float *wpk, int32_t cout, int32_t cin,
                                       int32_t for_dgrad, void *stream) {
    MVX_CHECK_ARG(w && wpk && cout > 0 && cin > 0);
    MVX_CHECK_ARG(((for_dgrad & 1) ? cout : cin) % BK == 0);
    const long long total = 27ll * cout * cin;
    hipLaunchKernelGGL(pack_weights, dim3(mvx_cdiv(total, 256) > 2048 ? 2048 : mvx_cdiv(total, 256)), dim3(256), 0,
                       (hipStream_t)stream, w, wpk, cout, cin, for_dgrad & 1, (for_dgrad >> 1) & 1);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

static int check_geom(int32_t din, int32_t dout, int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t sd,
                      int32_t pd) {
    if (din <= 0 || dout <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0) return MVX_EINVAL;
    if (sd < 1 || sd > 2 || pd < 0 || pd > 1) return MVX_EINVAL;
    if (cin % BK || cout % BN) return MVX_ESIZE;
    return MVX_OK;
}

// Launch the gather kernel: persistent form when the caller supplies a zeroed work counter, classic grid otherwise.
static int persistent_grid() {
    static int cached = 0;
    if (!cached) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            cus <= 0)
            cus = 256;
        cached = 2 * cus;                           // two workgroups per CU (three would fit: 52.7 KB of LDS, 158 VGPRs each)
    }
    return cached;
}

// Narrow units: a launch with fewer 64-channel units than this runs as twice as many 32-channel units (gather_unit<.., 1>),
// each half the matrix work -- when that lets ALL units be resident at once (three workgroups per CU): up to 1.5 units per
// CU.  Measured per RPN layer shape (tools/time_conv2d.py <frames> <limit>), together with the classic launch up to three
// units per CU (pf_max in launch_gather): four frames, 44 x 50 maps (384 -> 768 units) 0.165 -> 0.135 ms forward, 0.150 ->
// 0.119 input gradient, the stride-2 layer 0.119 -> 0.089; 88 x 100 (616 units, stay 64-channel) 0.144 -> 0.133; two frames
// 88 x 100 (308 -> 616) 0.092 -> 0.078; one frame 44 x 50 (96 -> 192) 0.092 -> 0.060.  Narrow units beyond that point
// (more units than fit at once) are slower: a unit is bound by the serial chain of its K stages, not by its MFMAs.
// Results are bit identical (the K order of an output element does not change).  Tuning value
// MVX_TUNE_GATHER_NARROW_MAX_UNITS (negative = this rule).
static long long g_gather_narrow_max_units = -1;
void mvxi_gather_narrow_max_units(long long v) { g_gather_narrow_max_units = v; }

static void launch_gather(hipStream_t st, const float *in, const float *wpk, const float *bias, float *out, double *stats,
                          const Geom &g, int relu, const int *in_hflag, const unsigned char *out_mask, const float *bg_pre,
                          int border_active, unsigned long long *exec_stages, const int *only_tiles, unsigned *done_counter,
                          double fin_count, double fin_eps, float *fin_mean_inv, unsigned *work_counter) {
    dim3 grid = gather_grid(g);
    const long long narrow_max = g_gather_narrow_max_units >= 0 ? g_gather_narrow_max_units : 3ll * (persistent_grid() / 2) / 2 + 1;
    const bool narrow = (long long)grid.x * grid.y * grid.z < narrow_max;
    if (narrow) grid.z *= 2;
    const long long units = (long long)grid.x * grid.y * grid.z;
    // one workgroup per unit while all units fit on the GPU at once (three per CU: 52.7 KB of LDS, <= 168 VGPRs): a persistent
    // grid of two per CU would run 513..768 units in two rounds
    const long long pf_max = 3ll * (persistent_grid() / 2);
#define MVX_LAUNCH_GATHER_N(TLO, THI, NB2)                                                                                       \
    do {                                                                                                                         \
        if (work_counter && units > pf_max)                                                                                      \
            hipLaunchKernelGGL((conv3d_gather_pw<TLO, THI, NB2>), dim3(persistent_grid()), dim3(256), 0, st, in, wpk, bias, out,  \
                               stats, g, relu, in_hflag, out_mask, bg_pre, border_active, exec_stages, only_tiles, done_counter,  \
                               fin_count, fin_eps, fin_mean_inv, work_counter, (int)grid.x, (int)grid.y, (int)grid.z);            \
        else                                                                                                                     \
            hipLaunchKernelGGL((conv3d_gather_pf<TLO, THI, NB2>), grid, dim3(256), 0, st, in, wpk, bias, out, stats, g, relu,     \
                               in_hflag, out_mask, bg_pre, border_active, exec_stages, only_tiles, done_counter, fin_count,       \
                               fin_eps, fin_mean_inv);                                                                            \
    } while (0)
#define MVX_LAUNCH_GATHER(TLO, THI)                                                                                              \
    do {                                                                                                                         \
        if (narrow) MVX_LAUNCH_GATHER_N(TLO, THI, 1);                                                                            \
        else MVX_LAUNCH_GATHER_N(TLO, THI, 2);                                                                                   \
    } while (0)
    if (g.tap_lo == 0 && g.tap_hi == 3) MVX_LAUNCH_GATHER(0, 3);
    else if (g.tap_lo == 0 && g.tap_hi == 2) MVX_LAUNCH_GATHER(0, 2);
    else MVX_LAUNCH_GATHER(1, 3);
#undef MVX_LAUNCH_GATHER
#undef MVX_LAUNCH_GATHER_N
}

extern "C" void mvx_conv3d_tile_shape(int32_t *tile_h, int32_t *tile_w) {
    if (tile_h) *tile_h = TH;
    if (tile_w) *tile_w = TW;
}

extern "C" int mvx_conv3d_forward(const float *in, const float *wpk, const float *bias, float *out,
                                  double *stats, int32_t din, int32_t dout, int32_t h, int32_t w,
                                  int32_t cin, int32_t cout, int32_t stride_d, int32_t pad_d,
                                  int32_t flags, uint32_t *work_counter, void *stream) {
    MVX_CHECK_ARG(in && wpk && out);
    int rc = check_geom(din, dout, h, w, cin, cout, stride_d, pad_d);
    if (rc) return rc;
    MVX_CHECK_ARG(dout == (din + 2 * pad_d - 3) / stride_d + 1);
    hipStream_t st = (hipStream_t)stream;
    const int relu = flags & MVX_FLAG_RELU;
    if (stats && !(flags & MVX_FLAG_PREZEROED)) {
        hipError_t e = hipMemsetAsync(stats, 0, sizeof(double) * MVX_REP * 2 * cout, st);
        if (e != hipSuccess) return (int)e;
    }
    Geom g{din, dout, h, w, cin, cout, stride_d, pad_d, 0};
    launch_gather(st, in, wpk, bias, out, stats, g, relu, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0.0, 0.0,
                  nullptr, (unsigned *)work_counter);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" int mvx_conv3d_forward_bg_frames(const float *in, const float *wpk, const float *bias, float *out, double *stats,
                                            int32_t din, int32_t dout, int32_t h, int32_t w, int32_t cin, int32_t cout,
                                            int32_t stride_d, int32_t pad_d, int32_t flags, const int32_t *in_halo_flags,
                                            const uint8_t *out_mask, const float *bg_pre, int32_t border_active,
                                            uint64_t *exec_stages, uint32_t *done_counter, double count, double eps,
                                            float *mean_inv, uint32_t *work_counter, int32_t n_frames, void *stream) {
    MVX_CHECK_ARG(in && wpk && out && in_halo_flags && out_mask && bg_pre);
    int rc = check_geom(din, dout, h, w, cin, cout, stride_d, pad_d);
    if (rc) return rc;
    MVX_CHECK_ARG(dout == (din + 2 * pad_d - 3) / stride_d + 1);
    MVX_CHECK_ARG(n_frames >= 1 && n_frames <= MVX_MAX_FRAMES);
    hipStream_t st = (hipStream_t)stream;
    if (stats && !(flags & MVX_FLAG_PREZEROED)) {
        hipError_t e = hipMemsetAsync(stats, 0, sizeof(double) * MVX_REP * 2 * cout * n_frames, st);
        if (e != hipSuccess) return (int)e;
    }
    if (done_counter) {
        MVX_CHECK_ARG(stats && mean_inv && count > 0);
        if (!(flags & MVX_FLAG_PREZEROED)) {
            hipError_t e = hipMemsetAsync(done_counter, 0, sizeof(uint32_t), st);
            if (e != hipSuccess) return (int)e;
        }
    }
    Geom g{din, dout, h, w, cin, cout, stride_d, pad_d, 0, n_frames};
    launch_gather(st, in, wpk, bias, out, stats, g, flags & MVX_FLAG_RELU, in_halo_flags, out_mask, bg_pre,
                  (border_active ? 1 : 0) | ((flags & MVX_FLAG_BG_TAPS) ? 2 : 0),
                  (unsigned long long *)exec_stages, nullptr, (unsigned *)done_counter, count, eps, mean_inv,
                  (unsigned *)work_counter);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" int mvx_conv3d_forward_bg(const float *in, const float *wpk, const float *bias, float *out, double *stats,
                                     int32_t din, int32_t dout, int32_t h, int32_t w, int32_t cin, int32_t cout,
                                     int32_t stride_d, int32_t pad_d, int32_t flags, const int32_t *in_halo_flags,
                                     const uint8_t *out_mask, const float *bg_pre, int32_t border_active,
                                     uint64_t *exec_stages, uint32_t *done_counter, double count, double eps,
                                     float *mean_inv, uint32_t *work_counter, void *stream) {
    return mvx_conv3d_forward_bg_frames(in, wpk, bias, out, stats, din, dout, h, w, cin, cout, stride_d, pad_d, flags,
                                        in_halo_flags, out_mask, bg_pre, border_active, exec_stages, done_counter, count, eps,
                                        mean_inv, work_counter, 1, stream);
}

static int launch_dgrad(const float *dz, const float *wpk_dgrad, float *dx, int32_t din, int32_t dout, int32_t h,
                        int32_t w, int32_t cin, int32_t cout, int32_t stride_d, int32_t pad_d, const int32_t *only_tiles,
                        uint64_t *exec_stages, uint32_t *work_counter, void *stream, int32_t n_frames = 1) {
    MVX_CHECK_ARG(dz && wpk_dgrad && dx);
    MVX_CHECK_ARG(n_frames >= 1 && n_frames <= MVX_MAX_FRAMES);
    // gather view: source = dz (dout planes, cout channels), result = dx (din planes, cin channels)
    int rc = check_geom(din, dout, h, w, cout, cin, stride_d, pad_d);
    if (rc) return rc;
    Geom g{dout, din, h, w, cout, cin, stride_d, pad_d, 1, n_frames};
    launch_gather((hipStream_t)stream, dz, wpk_dgrad, nullptr, dx, nullptr, g, 0, nullptr, nullptr, nullptr, 0,
                  (unsigned long long *)exec_stages, only_tiles, nullptr, 0.0, 0.0, nullptr, (unsigned *)work_counter);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" int mvx_conv3d_dgrad(const float *dz, const float *wpk_dgrad, float *dx, int32_t din,
                                int32_t dout, int32_t h, int32_t w, int32_t cin, int32_t cout,
                                int32_t stride_d, int32_t pad_d, uint32_t *work_counter, void *stream) {
    return launch_dgrad(dz, wpk_dgrad, dx, din, dout, h, w, cin, cout, stride_d, pad_d, nullptr, nullptr, work_counter, stream);
}

extern "C" int mvx_conv3d_dgrad_tiles(const float *dz, const float *wpk_dgrad, float *dx, int32_t din, int32_t dout,
                                      int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t stride_d, int32_t pad_d,
                                      const int32_t *dx_tile_flags, uint64_t *exec_stages, uint32_t *work_counter,
                                      void *stream) {
    MVX_CHECK_ARG(dx_tile_flags);
    return launch_dgrad(dz, wpk_dgrad, dx, din, dout, h, w, cin, cout, stride_d, pad_d, dx_tile_flags, exec_stages,
                        work_counter, stream);
}

extern "C" int mvx_conv3d_dgrad_tiles_frames(const float *dz, const float *wpk_dgrad, float *dx, int32_t din, int32_t dout,
                                             int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t stride_d, int32_t pad_d,
                                             const int32_t *dx_tile_flags, uint64_t *exec_stages, uint32_t *work_counter,
                                             int32_t n_frames, void *stream) {
    MVX_CHECK_ARG(dx_tile_flags);
    return launch_dgrad(dz, wpk_dgrad, dx, din, dout, h, w, cin, cout, stride_d, pad_d, dx_tile_flags, exec_stages,
                        work_counter, stream, n_frames);
}

static int wgrad_strips(int h, int w, int cin) {
    // Two workgroups fit a CU (LDS), so 512 run at once; all have the same length, so the grid
    // should fill exactly one round: strips x (3 depth taps x channel chunks) <= 512.
    const int ntiles = (int)(mvx_cdiv(w, TW) * mvx_cdiv(h, TH));
    const int chunks = (cin % W4_C == 0) ? cin / W4_C : cin / BK;
    int strips = 512 / (3 * chunks);
    if (strips < 1) strips = 1;
    int per = (ntiles + strips - 1) / strips;
    if (per < 1) per = 1;
    return per;
}

extern "C" size_t mvx_conv3d_wgrad_workspace_bytes(int32_t h, int32_t w, int32_t cin, int32_t cout) {
    if (h <= 0 || w <= 0 || cin <= 0 || cout <= 0 || cout % BN) return 0;
    const int ntiles = (int)(mvx_cdiv(w, TW) * mvx_cdiv(h, TH));
    const int per = wgrad_strips(h, w, cin);
    const int nstrips = (ntiles + per - 1) / per;
    return (size_t)(cout / BN) * nstrips * 27 * cin * BN * sizeof(float);
}

extern "C" int mvx_conv3d_wgrad(const float *in, const float *dz, float *dw, int32_t din, int32_t dout,
                                int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t stride_d,
                                int32_t pad_d, int32_t flags, void *workspace, size_t workspace_bytes, void *stream) {
    const SplitAmax am = mvxi_take_split_amax();         // (in, dz) bound for this call (fp16 pieces); cleared whatever kernel runs
    MVX_CHECK_ARG(in && dz && dw && workspace);
    int rc = check_geom(din, dout, h, w, cin, cout, stride_d, pad_d);
    if (rc) return rc;
    const int nblk = cout / BN;                     // the 4-wave kernel takes any multiple of 64 output channels
    if (cout != BN && cin % W4_C) return MVX_ESIZE;
    const int ntiles = (int)(mvx_cdiv(w, TW) * mvx_cdiv(h, TH));
    const int per = wgrad_strips(h, w, cin);
    const int nstrips = (ntiles + per - 1) / per;
    MVX_CHECK_ARG(workspace_bytes >= (size_t)nblk * nstrips * 27 * cin * BN * sizeof(float));
    Geom g{din, dout, h, w, cin, cout, stride_d, pad_d, 0};
    hipStream_t st = (hipStream_t)stream;
    if (cin % W4_C == 0)
        launch_wgrad4<false>(flags, dim3(nstrips, 3 * (cin / W4_C), nblk), st, in, dz, (float *)workspace, g, per, nullptr, nullptr,
                             nullptr, Strips{{nstrips, nstrips, nstrips}}, am);
    else
        hipLaunchKernelGGL(conv3d_wgrad, dim3(nstrips, 3 * (cin / BK)), dim3(WG_THREADS), 0, st, in, dz,
                           (float *)workspace, g, per);
    MVX_LAUNCH_CHECK();
    const size_t per_slab = (size_t)27 * cin * BN;
    hipLaunchKernelGGL(wgrad_reduce, dim3(mvx_cdiv(per_slab, 256), nblk), dim3(256), 0, st, (const float *)workspace, dw,
                       nstrips, cin, flags & MVX_FLAG_ACCUMULATE, (flags & MVX_FLAG_CONV2D) ? 1 : 0,
                       Strips{{nstrips, nstrips, nstrips}});
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

// mvx_conv3d_wgrad_bg: one workgroup per CU (the kernel's 9 accumulator tiles leave room for one wave per SIMD),
// strips x 3 depth taps x channel chunks = 255..256 workgroups, steps dealt round-robin from the compacted lists.
// workspace: [slabs][R f64 replicas][T][step lists][step counts]
static int wgrad_bg_strips(int cin) {
    // slab capacity per depth tap: up to half of the 256 workgroups of a chunk set (the busiest tap of a stride-2
    // layer has twice the planes of the others)
    const int s = 128 / (cin / W4_C);
    return s < 1 ? 1 : s;
}
static size_t wgrad_bg_slab_bytes(int cin) { return (size_t)wgrad_bg_strips(cin) * 27 * cin * BN * sizeof(float); }

int mvxi_wgrad_step_list(const int32_t *in_halo_flags, int din, int dout, int ntiles, int stride_d, int pad_d, int *list,
                         int *count, hipStream_t st, int n_frames) {
    Geom g{din, dout, 0, 0, 0, 0, stride_d, pad_d, 0, n_frames};
    hipLaunchKernelGGL(wgrad_step_list, dim3(3), dim3(1024), 0, st, in_halo_flags, g, ntiles, list, count);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

int mvxi_wgrad_rank1(const float *tap_sums, const float *c_in, float *dw, int din, int dout, int cin, int cout, int stride_d,
                     int pad_d, hipStream_t st, int n_frames) {
    Geom g{din, dout, 0, 0, cin, cout, stride_d, pad_d, 0, n_frames};
    const size_t total = (size_t)27 * cin * cout;
    hipLaunchKernelGGL(wgrad_rank1, dim3(mvx_cdiv(total, 256)), dim3(256), 0, st, tap_sums, c_in, dw, g);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" size_t mvx_plane_tap_sums_workspace_bytes(int32_t planes, int32_t channels) {
    return planes > 0 && channels > 0 ? sizeof(double) * RREP * planes * RK * channels : 0;
}

extern "C" int mvx_plane_tap_sums(const float *dz, int32_t planes, int32_t h, int32_t w, int32_t channels,
                                  const int32_t *tile_flags, const float *inactive_sums, float *tap_sums,
                                  void *workspace, size_t workspace_bytes, void *stream) {
    MVX_CHECK_ARG(dz && tap_sums && workspace && planes > 0 && h > 0 && w > 0 && channels > 0);
    MVX_CHECK_ARG(channels % 4 == 0 && 256 % (channels / 4) == 0);
    MVX_CHECK_ARG((tile_flags == nullptr) == (inactive_sums == nullptr));
    MVX_CHECK_ARG(workspace_bytes >= mvx_plane_tap_sums_workspace_bytes(planes, channels));
    hipStream_t st = (hipStream_t)stream;
    double *R = (double *)workspace;
    hipError_t e = hipMemsetAsync(R, 0, sizeof(double) * RREP * planes * RK * channels, st);
    if (e != hipSuccess) return (int)e;
    if (tile_flags)
        hipLaunchKernelGGL(plane_region_sums_tiles, dim3(mvx_cdiv(w, TW) * mvx_cdiv(h, TH), planes), dim3(256), 0, st, dz,
                           tile_flags, planes, h, w, channels, R);
    else
        hipLaunchKernelGGL(plane_region_sums, dim3(h, planes), dim3(256), 0, st, dz, planes, h, w, channels, R);
    MVX_LAUNCH_CHECK();
    hipLaunchKernelGGL(region_tap_sums, dim3(mvx_cdiv((long long)planes * 9 * channels, 64)), dim3(64), 0, st,
                       (const double *)R, inactive_sums, planes, channels, tap_sums);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" int mvx_conv3d_input_grad_sums_frames(const float *w, const float *tap_sums, int32_t din, int32_t dout, int32_t cin,
                                                 int32_t cout, int32_t stride_d, int32_t pad_d, float *plane_grad_sums,
                                                 int32_t n_frames, void *stream) {
    MVX_CHECK_ARG(w && tap_sums && plane_grad_sums && din > 0 && dout > 0 && cin > 0 && cout > 0);
    MVX_CHECK_ARG(n_frames >= 1 && n_frames <= MVX_MAX_FRAMES);
    Geom g{din, dout, 0, 0, cin, cout, stride_d, pad_d, 0, n_frames};
    hipLaunchKernelGGL(input_grad_sums, dim3(cin, din * n_frames), dim3(64), 0, (hipStream_t)stream, w, tap_sums, g,
                       plane_grad_sums);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" int mvx_conv3d_input_grad_sums(const float *w, const float *tap_sums, int32_t din, int32_t dout, int32_t cin,
                                          int32_t cout, int32_t stride_d, int32_t pad_d, float *plane_grad_sums,
                                          void *stream) {
    return mvx_conv3d_input_grad_sums_frames(w, tap_sums, din, dout, cin, cout, stride_d, pad_d, plane_grad_sums, 1, stream);
}

extern "C" size_t mvx_conv3d_wgrad_bg_workspace_bytes_frames(int32_t dout, int32_t h, int32_t w, int32_t cin, int32_t cout,
                                                            int32_t n_frames) {
    if (dout <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout != BN || cin % W4_C || n_frames <= 0) return 0;
    const size_t ntiles = (size_t)mvx_cdiv(w, TW) * mvx_cdiv(h, TH);
    return wgrad_bg_slab_bytes(cin) + sizeof(int) * (3 * (size_t)dout * n_frames * ntiles + 4);
}

extern "C" size_t mvx_conv3d_wgrad_bg_workspace_bytes(int32_t dout, int32_t h, int32_t w, int32_t cin, int32_t cout) {
    return mvx_conv3d_wgrad_bg_workspace_bytes_frames(dout, h, w, cin, cout, 1);
}

extern "C" int mvx_conv3d_wgrad_bg_frames(const float *in, const float *dz, float *dw, int32_t din, int32_t dout, int32_t h,
                                          int32_t w, int32_t cin, int32_t cout, int32_t stride_d, int32_t pad_d, int32_t flags,
                                          const int32_t *in_halo_flags, const float *c_in, const float *tap_sums,
                                          void *workspace, size_t workspace_bytes, int32_t n_frames, void *stream) {
    const SplitAmax am = mvxi_take_split_amax();         // (in, dz) bound for this call (fp16 pieces); cleared whatever kernel runs
    MVX_CHECK_ARG(in && dz && dw && workspace && in_halo_flags && c_in && tap_sums);
    int rc = check_geom(din, dout, h, w, cin, cout, stride_d, pad_d);
    if (rc) return rc;
    if (cout != BN || cin % W4_C) return MVX_ESIZE;
    MVX_CHECK_ARG(n_frames >= 1 && n_frames <= MVX_MAX_FRAMES);
    MVX_CHECK_ARG(workspace_bytes >= mvx_conv3d_wgrad_bg_workspace_bytes_frames(dout, h, w, cin, cout, n_frames));
    const int ntiles = (int)(mvx_cdiv(w, TW) * mvx_cdiv(h, TH));
    const int nstrips = wgrad_bg_strips(cin);       // slab capacity = the largest share a depth tap can get
    Geom g{din, dout, h, w, cin, cout, stride_d, pad_d, 0, n_frames};
    hipStream_t st = (hipStream_t)stream;
    // workgroups per depth tap in proportion to its valid output planes (conv3: 1 / 2 / 1 of 2 planes), 256 per chunk set
    int nd[3], ndsum = 0;
    for (int kd = 0; kd < 3; ++kd) {
        nd[kd] = 0;
        for (int d = 0; d < dout; ++d) {
            const int ds = d * stride_d - pad_d + kd;
            nd[kd] += ds >= 0 && ds < din;
        }
        ndsum += nd[kd];
    }
    Strips ks;
    int widest = 1;
    for (int kd = 0; kd < 3; ++kd) {
        int share = ndsum > 0 ? (2 * nstrips * nd[kd]) / ndsum : nstrips;      // 2 * nstrips = 256 / chunks workgroups in all
        if (share < 1) share = 1;
        if (share > nstrips) share = nstrips;       // never beyond the slab capacity
        ks.n[kd] = share;
        if (share > widest) widest = share;
    }
    float *slabs = (float *)workspace;
    int *list = (int *)((char *)workspace + wgrad_bg_slab_bytes(cin));
    int *count = list + (size_t)3 * dout * n_frames * ntiles;
    hipLaunchKernelGGL(wgrad_step_list, dim3(3), dim3(1024), 0, st, in_halo_flags, g, ntiles, list, count);
    MVX_LAUNCH_CHECK();
    launch_wgrad4<false>(flags, dim3(widest, 3 * (cin / W4_C)), st, in, dz, slabs, g, 0, list, count, c_in, ks, am);
    MVX_LAUNCH_CHECK();
    const size_t per_slab = (size_t)27 * cin * BN;
    hipLaunchKernelGGL(wgrad_reduce, dim3(mvx_cdiv(per_slab, 256)), dim3(256), 0, st, (const float *)slabs, dw, widest, cin,
                       flags & MVX_FLAG_ACCUMULATE, 0, ks);
    MVX_LAUNCH_CHECK();
    hipLaunchKernelGGL(wgrad_rank1, dim3(mvx_cdiv(per_slab, 256)), dim3(256), 0, st, tap_sums, c_in, dw, g);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" int mvx_conv3d_wgrad_bg(const float *in, const float *dz, float *dw, int32_t din, int32_t dout, int32_t h,
                                   int32_t w, int32_t cin, int32_t cout, int32_t stride_d, int32_t pad_d, int32_t flags,
                                   const int32_t *in_halo_flags, const float *c_in, const float *tap_sums,
                                   void *workspace, size_t workspace_bytes, void *stream) {
    return mvx_conv3d_wgrad_bg_frames(in, dz, dw, din, dout, h, w, cin, cout, stride_d, pad_d, flags, in_halo_flags, c_in,
                                      tap_sums, workspace, workspace_bytes, 1, stream);
}

// ------------------------------------------------------------------------------------------
// 2-D convolutions of the RPN on frame sets (modules/voxelnet/Pipe.py:45-75): 3x3, stride 1, padding 1 on
// channels-last [F][H][W][C] maps = the gather / wgrad kernels above with one plane per frame (depth tap 1 only).
// MVX_FLAG_TAPS2: only the 2x2 window of taps {0,1}^2 carries weight (a stride-2 3x3 convolution evaluated on the
// space-to-depth image of its input); the dgrad of such a layer reads the flipped window {1,2}^2.
// ------------------------------------------------------------------------------------------
static int conv2d_geom_ok(int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t n_frames) {
    if (h <= 0 || w <= 0 || cin <= 0 || cout <= 0 || n_frames < 1 || n_frames > MVX_MAX_FRAMES) return MVX_EINVAL;
    if (cin % BK || cout % BN) return MVX_ESIZE;
    return MVX_OK;
}

extern "C" int mvx_conv2d_forward_frames(const float *in, const float *wpk, const float *bias, float *out, double *stats,
                                         int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t flags,
                                         uint32_t *done_counter, double eps, float *mean_inv, uint32_t *work_counter,
                                         int32_t n_frames, void *stream) {
    MVX_CHECK_ARG(in && wpk && out);
    int rc = conv2d_geom_ok(h, w, cin, cout, n_frames);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (stats && !(flags & MVX_FLAG_PREZEROED)) {
        hipError_t e = hipMemsetAsync(stats, 0, sizeof(double) * MVX_REP * 2 * cout * n_frames, st);
        if (e != hipSuccess) return (int)e;
    }
    if (done_counter) {
        MVX_CHECK_ARG(stats && mean_inv);
        if (!(flags & MVX_FLAG_PREZEROED)) {
            hipError_t e = hipMemsetAsync(done_counter, 0, sizeof(uint32_t), st);
            if (e != hipSuccess) return (int)e;
        }
    }
    Geom g{1, 1, h, w, cin, cout, 1, 1, 0, n_frames, 0, (flags & MVX_FLAG_TAPS2) ? 2 : 3};
    // the structurally zero (window tap, parity) blocks of the rearranged stride-2 kernel are not executed (7 of 16), when
    // the parity blocks are whole K chunks
    if ((flags & MVX_FLAG_TAPS2) && cin % 4 == 0 && (cin / 4) % BK == 0) g.s2d = cin / 4;
    launch_gather(st, in, wpk, bias, out, stats, g, flags & MVX_FLAG_RELU, nullptr, nullptr, nullptr, 0, nullptr, nullptr,
                  (unsigned *)done_counter, (double)h * w, eps, mean_inv, (unsigned *)work_counter);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" int mvx_conv2d_dgrad_frames(const float *dz, const float *wpk_dgrad, float *dx, int32_t h, int32_t w, int32_t cin,
                                       int32_t cout, int32_t flags, uint32_t *work_counter, int32_t n_frames, void *stream) {
    MVX_CHECK_ARG(dz && wpk_dgrad && dx);
    int rc = conv2d_geom_ok(h, w, cout, cin, n_frames);          // gather view: source dz (cout channels) -> dx (cin channels)
    if (rc) return rc;
    Geom g{1, 1, h, w, cout, cin, 1, 1, 1, n_frames, (flags & MVX_FLAG_TAPS2) ? 1 : 0, 3};
    if ((flags & MVX_FLAG_TAPS2) && cin % 4 == 0 && (cin / 4) % BN == 0) g.s2d = cin / 4;      // parity of the OUTPUT channel block
    launch_gather((hipStream_t)stream, dz, wpk_dgrad, nullptr, dx, nullptr, g, 0, nullptr, nullptr, nullptr, 0, nullptr, nullptr,
                  nullptr, 0.0, 0.0, nullptr, (unsigned *)work_counter);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

static int conv2d_wgrad_strips(int cin, int cout) {
    int s = 256 / ((cin / W4_C) * (cout / BN));                   // one 8-wave workgroup per CU in all (slab capacity)
    if (s > 128) s = 128;
    return s < 1 ? 1 : s;
}

extern "C" size_t mvx_conv2d_wgrad_workspace_bytes_frames(int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t n_frames) {
    if (h <= 0 || w <= 0 || cin <= 0 || cout <= 0 || cin % W4_C || cout % BN || n_frames <= 0) return 0;
    const size_t ntiles = (size_t)mvx_cdiv(w, TW) * mvx_cdiv(h, TH);
    const size_t slabs = (size_t)(cout / BN) * conv2d_wgrad_strips(cin, cout) * 27 * cin * BN * sizeof(float);
    return slabs + sizeof(int) * (3 * (size_t)n_frames * ntiles + 4) + sizeof(int) * (size_t)n_frames * ntiles;
}

// dw f32 [cout][cin][3][3] (ADDED to with MVX_FLAG_ACCUMULATE) = sum over all frames and sites of in (x) dz
extern "C" int mvx_conv2d_wgrad_frames(const float *in, const float *dz, float *dw, int32_t h, int32_t w, int32_t cin,
                                       int32_t cout, int32_t flags, void *workspace, size_t workspace_bytes, int32_t n_frames,
                                       void *stream) {
    const SplitAmax am = mvxi_take_split_amax();         // (in, dz) bound for this call (fp16 pieces); cleared whatever kernel runs
    MVX_CHECK_ARG(in && dz && dw && workspace);
    int rc = conv2d_geom_ok(h, w, cin, cout, n_frames);
    if (rc) return rc;
    if (cin % W4_C) return MVX_ESIZE;
    MVX_CHECK_ARG(workspace_bytes >= mvx_conv2d_wgrad_workspace_bytes_frames(h, w, cin, cout, n_frames));
    hipStream_t st = (hipStream_t)stream;
    const int ntiles = (int)(mvx_cdiv(w, TW) * mvx_cdiv(h, TH));
    int nstrips = conv2d_wgrad_strips(cin, cout);
    if (nstrips > (n_frames * ntiles + 3) / 4) nstrips = (n_frames * ntiles + 3) / 4;     // >= 4 tile steps per strip: the slabs
    if (nstrips < 1) nstrips = 1;                                                       // (589 KB per strip at cin 128) stay small
    const int nblk = cout / BN;
    Geom g{1, 1, h, w, cin, cout, 1, 1, 0, n_frames, 0, (flags & MVX_FLAG_TAPS2) ? 2 : 3};
    if ((flags & MVX_FLAG_TAPS2) && cin % 4 == 0 && (cin / 4) % W4_C == 0) g.s2d = cin / 4;
    float *slabs = (float *)workspace;
    int *list = (int *)((char *)workspace + (size_t)nblk * nstrips * 27 * cin * BN * sizeof(float));
    int *count = list + (size_t)3 * n_frames * ntiles;
    int *ones = count + 4;                                         // "every tile is a step": any non-zero word is a set flag
    hipError_t e = hipMemsetAsync(ones, 0x01, sizeof(int) * (size_t)n_frames * ntiles, st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(wgrad_step_list, dim3(3), dim3(1024), 0, st, (const int *)ones, g, ntiles, list, count);
    MVX_LAUNCH_CHECK();
    Strips ks;
    ks.n[0] = 1; ks.n[1] = nstrips; ks.n[2] = 1;                    // depth tap 1 is the only one with a source plane
    if (flags & MVX_FLAG_TAPS2)
        launch_wgrad4<true>(flags, dim3(nstrips, 3 * (cin / W4_C), nblk), st, in, dz, slabs, g, 0, list, count, nullptr, ks, am);
    else
        launch_wgrad4<false>(flags, dim3(nstrips, 3 * (cin / W4_C), nblk), st, in, dz, slabs, g, 0, list, count, nullptr, ks, am);
    MVX_LAUNCH_CHECK();
    const size_t per_slab = (size_t)27 * cin * BN;
    hipLaunchKernelGGL(wgrad_reduce, dim3(mvx_cdiv(per_slab, 256), nblk), dim3(256), 0, st, (const float *)slabs, dw, nstrips, cin,
                       flags & MVX_FLAG_ACCUMULATE, 1, ks);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}
