// Knock-out probe of conv3d_gather_pf: the production kernel text with pieces removed by template knobs
// (results are garbage; only the time matters).  Snapshot of the ROUND-1 kernel text (generated then by a script that tracked the kernel source; the production kernel has since gained the frame dimension, the tap-range template and the swizzled weight tile).
//   knob 1: no barriers between tap rows   2: no weight traffic (global loads + LDS stores)
//        4: no halo traffic                8: MFMA operands from registers (no LDS reads)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define MVX_REP 32
static inline unsigned mvx_cdiv(long long a, long long b) { return (unsigned)((a + b - 1) / b); }
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int TH = 8, TW = 16, HH = TH + 2, HW = TW + 2, BK = 32, PITCH = BK + 4, BN = 64;
struct Geom { int Din, Dout, H, W, Cin, Cout, sd, pd, mode; };
__device__ __forceinline__ int src_depth(const Geom &g, int d, int kd) {
    if (g.mode == 0) { const int s = d * g.sd - g.pd + kd; return (s >= 0 && s < g.Din) ? s : -1; }
    const int t = d + g.pd - kd;
    if (t < 0 || (t % g.sd) != 0) return -1;
    const int s = t / g.sd;
    return s < g.Din ? s : -1;
}
namespace {
constexpr int WROW = BN * PITCH;           // one tap's weight tile in LDS
// Halo row stride: a multiple of 64 floats, so that the two patch rows a wave reads (lanes 0-15 / 16-31)
// start on the same 16-byte slot of the 256-byte bank row.  ds_read_b128 is served in the lane groups
// {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32): with site pitch 36 floats (9 slots, odd) the slots of one
// row are a permutation of 0..15, and equal row phases make every group hit 16 distinct slots.
constexpr int HROW = ((HW * PITCH + 63) / 64) * 64;

// Launch geometry: (tiles, output planes, 64-channel blocks).  An XCD-contiguous remap of the (tile, plane)
// space was measured and was not faster (forward equal, stride-2 dgrad slower), so the plain grid stays.
inline dim3 gather_grid(const Geom &g) { return dim3(mvx_cdiv(g.W, TW) * mvx_cdiv(g.H, TH), g.Dout, g.Cout / BN); }

template <int KNOB>
__global__ __launch_bounds__(256, 2) void conv3d_gather_pf(const float *__restrict__ in,
                                                           const float *__restrict__ wpk,
                                                           const float *__restrict__ bias,
                                                           float *__restrict__ out, double *__restrict__ stats,
                                                           Geom g, int relu) {
    __shared__ __attribute__((aligned(16))) float s_halo[HH * HROW];
    __shared__ __attribute__((aligned(16))) float s_w[3 * WROW];
    __shared__ float s_red[4][2 * BN];

    const int tiles_x = (g.W + TW - 1) / TW;
    const int tx0 = (blockIdx.x % tiles_x) * TW, ty0 = (blockIdx.x / tiles_x) * TH;
    const int d = blockIdx.y, nb = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int nchunks = g.Cin / BK;

    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }

    const int my_ty = 2 * wv + (li >> 4), my_tx = li & 15;
    const int a_base = my_ty * HROW + my_tx * PITCH + 4 * lh;
    const int b_base0 = li * PITCH + 4 * lh;
    const int b_base1 = (32 + li) * PITCH + 4 * lh;

    // valid depth taps of this output plane (block-uniform), packed as (kd, source plane) pairs
    int kd_l[3] = {0, 0, 0}, ds_l[3] = {0, 0, 0}, nk = 0;
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) {
        const int ds = src_depth(g, d, kd);
        if (ds >= 0) {
            if (nk == 0) { kd_l[0] = kd; ds_l[0] = ds; }
            else if (nk == 1) { kd_l[1] = kd; ds_l[1] = ds; }
            else { kd_l[2] = kd; ds_l[2] = ds; }
            ++nk;
        }
    }
    const int nstages = nk * nchunks;
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
        const float *row0 = wpk + ((((size_t)kd * 9 + row * 3) * nchunks + cc) * g.Cout + (size_t)nb * BN) * BK + (size_t)tid * 4;
        const size_t tap_stride = (size_t)nchunks * g.Cout * BK;
#pragma unroll
        for (int v = 0; v < 6; ++v) wreg[v] = *(const f32x4 *)(row0 + (v >> 1) * tap_stride + (v & 1) * 1024);
    };
    auto store_wrow = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int v = 0; v < 6; ++v) {
            const int c = tid + 256 * (v & 1);
            *(f32x4 *)(s_w + (v >> 1) * WROW + (c >> 3) * PITCH + (c & 7) * 4) = wreg[v];
        }
    };
    auto compute_row = [&](int row) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int a_off = a_base + row * HROW + t * PITCH;
#pragma unroll
            for (int q = 0; q < BK / 8; ++q) {
                float4 av, b0, b1;
                if (KNOB & 8) { av = make_float4(acc0[0], acc0[1], acc0[2], acc0[3]); b0 = make_float4(acc1[0], acc1[1], acc1[2], acc1[3]); b1 = b0; }
                else {
                    av = *(const float4 *)(s_halo + a_off + 8 * q);
                    b0 = *(const float4 *)(s_w + t * WROW + b_base0 + 8 * q);
                    b1 = *(const float4 *)(s_w + t * WROW + b_base1 + 8 * q);
                }
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
    };

    // Prefetches are unconditional (the last stage re-fetches its own operands and drops them): a
    // conditionally written register array would be demoted to scratch memory.
    load_wrow(0, 0);
    load_halo(0);
    for (int st = 0; st < nstages; ++st) {
        const int nxt = st + 1 < nstages ? st + 1 : st;
        __syncthreads();                           // previous stage's LDS reads are done
        if (!(KNOB & 4)) store_halo();
        if (!(KNOB & 2)) store_wrow();                              // tap row 0
        __syncthreads();
        if (!(KNOB & 2)) load_wrow(st, 1);                          // next weight row first ...
        if (!(KNOB & 4)) load_halo(nxt);                            // ... then the long-latency halo of the next stage
        compute_row(0);
        if (!(KNOB & 1)) __syncthreads();
        if (!(KNOB & 2)) store_wrow();                              // tap row 1 (waits for its 6 loads only)
        if (!(KNOB & 1)) __syncthreads();
        if (!(KNOB & 2)) load_wrow(st, 2);
        compute_row(1);
        if (!(KNOB & 1)) __syncthreads();
        if (!(KNOB & 2)) store_wrow();                              // tap row 2
        if (!(KNOB & 1)) __syncthreads();
        if (!(KNOB & 2)) load_wrow(nxt, 0);
        compute_row(2);
    }

    // ---- epilogue: bias, ReLU, store, BatchNorm statistics (identical to conv3d_gather)
    const int n0 = nb * BN + li, n1 = n0 + 32;
    const float bias0 = bias ? bias[n0] : 0.f, bias1 = bias ? bias[n1] : 0.f;
    float s1a = 0.f, s2a = 0.f, s1b = 0.f, s2b = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int gy = ty0 + 2 * wv + (row >> 4), gx = tx0 + (row & 15);
        float v0 = acc0[r] + bias0, v1 = acc1[r] + bias1;
        if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
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
            atomicAdd(stats + ((size_t)rep * 2 + which) * g.Cout + nb * BN + c, t);
        }
    }
}


}

template <int KNOB> void run(const char *what, float *in, float *w, float *b, float *out, Geom g) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid = gather_grid(g);
    conv3d_gather_pf<KNOB><<<grid, 256>>>(in, w, b, out, nullptr, g, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) conv3d_gather_pf<KNOB><<<grid, 256>>>(in, w, b, out, nullptr, g, 1);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double fl = 2.0 * 9 * 9 * g.H * g.W * 64.0 * 64.0;
    printf("knob %2d  %-44s %.3f ms  %.1f TFLOP/s-equivalent\n", KNOB, what, ms, fl / ms / 1e9);
}
__global__ void fill_random(float *p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (float)(h >> 8) * (1.0f / 16777216.0f) - 0.5f;
    }
}
int main(int argc, char **argv) {
    Geom g{5, 3, 352, 400, 64, 64, 1, 0, 0};
    float *in, *w, *b, *out;
    hipMalloc(&in, (size_t)5 * 352 * 400 * 64 * 4); hipMalloc(&out, (size_t)3 * 352 * 400 * 64 * 4);
    hipMalloc(&w, 27 * 64 * 64 * 4); hipMalloc(&b, 256);
    hipMemset(in, 0, (size_t)5 * 352 * 400 * 64 * 4); hipMemset(w, 0, 27 * 64 * 64 * 4); hipMemset(b, 0, 256);
    if (argc > 1) {
        fill_random<<<4096, 256>>>(in, (size_t)5 * 352 * 400 * 64, 1u);
        fill_random<<<256, 256>>>(w, 27 * 64 * 64, 2u);
        printf("random data\n");
    }
    run<0>("production", in, w, b, out, g);
    run<1>("no row barriers", in, w, b, out, g);
    run<2>("no weight traffic", in, w, b, out, g);
    run<4>("no halo traffic", in, w, b, out, g);
    run<6>("no weight, no halo traffic", in, w, b, out, g);
    run<7>("no weight/halo traffic, no row barriers", in, w, b, out, g);
    run<8>("no LDS reads", in, w, b, out, g);
    run<15>("MFMA only + stage barriers", in, w, b, out, g);
    return 0;
}
