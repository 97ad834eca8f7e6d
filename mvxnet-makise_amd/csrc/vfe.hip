// Voxel Feature Encoding glue: BatchNorm + per-voxel max over the T sampled rows + concat.
//
// Replaces, for one VFE layer (modules/voxelnet/Pipe.py:12-18):
//     x = BN(relu(fc(x)));  s = max_t x;  out = cat([x, repeat(s, T)], -1)
// and for the head (modules/voxelnet/VoxelNet.py:28-33):  x = BN(relu(fc(x)));  feat = max_t x
// The max runs over ALL T rows, padded ones included (no mask).  BatchNorm is a
// per-channel increasing affine map, so max_t BN(y) = BN(max_t y): the kernels normalise while
// they reduce and never materialise the intermediate tensor separately.
// Rows are [V][T][C] row-major; one thread owns one (voxel, channel) column, so loads and
// stores are coalesced across channels.
//
// Compact rows.  Inside MVXNet every padded row of a voxel is identical, so the T rows
// of voxel v can be stored as its vcnt[v] real rows (rows voff[v] .. voff[v]+vcnt[v]-1 of the
// matrix) plus ONE padded row (row n_real + v) that stands for the T - vcnt[v] identical padded
// rows.  With voff/vcnt given, the kernels below walk that layout; the padded row takes part in
// the max only if T - vcnt[v] > 0, and its gradient is the SUM over the rows it stands for.
// argmax holds the local row (0..vcnt-1) or vcnt[v] for the padded row.
#include "common.h"

namespace {

struct Rows {            // row addressing of one voxel
    const int *voff, *vcnt;
    int T, n_real;
    __device__ __forceinline__ int count(int v) const { return vcnt ? vcnt[v] : T; }
    __device__ __forceinline__ bool has_pad(int v) const { return vcnt && vcnt[v] < T; }
    // row index of local row t (t == count -> the padded row)
    __device__ __forceinline__ size_t row(int v, int t) const {
        if (!vcnt) return (size_t)v * T + t;
        return t < vcnt[v] ? (size_t)voff[v] + t : (size_t)n_real + v;
    }
};

__global__ __launch_bounds__(256) void vfe_bn_max_concat(const float *__restrict__ y, const float *__restrict__ mi,
                                                         float *__restrict__ out, int *__restrict__ argmax,
                                                         int V, int C, Rows R, FrameMap fm) {
    const long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (e >= (long long)V * C) return;
    const int v = (int)(e / C), c = (int)(e % C);
    mi += (size_t)fm_frame_of(fm, v) * 2 * C;             // per-frame BatchNorm (frames = voxel segments)
    const float m = mi[c], iv = mi[C + c];
    const int n = R.count(v) + (R.has_pad(v) ? 1 : 0);
    float best = -INFINITY;
    int bi = 0;
    for (int t = 0; t < n; ++t) {
        const float val = (y[R.row(v, t) * C + c] - m) * iv;
        if (val > best) { best = val; bi = t; }
    }
    const int nw = R.vcnt ? R.count(v) + 1 : n;          // the padded row is always written
    for (int t = 0; t < nw; ++t) {
        const size_t r = R.row(v, t);
        out[r * 2 * C + c] = (y[r * C + c] - m) * iv;
        out[r * 2 * C + C + c] = best;
    }
    argmax[e] = bi;
}

__global__ __launch_bounds__(256) void vfe_max_concat_bwd(const float *__restrict__ g, const int *__restrict__ argmax,
                                                          float *__restrict__ dyh, int V, int C, Rows R) {
    const long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (e >= (long long)V * C) return;
    const int v = (int)(e / C), c = (int)(e % C);
    const int nw = R.vcnt ? R.count(v) + 1 : R.T;        // rows stored for this voxel
    float s = 0.f;
    for (int t = 0; t < nw; ++t) s += g[R.row(v, t) * 2 * C + C + c];   // padded row: already summed
    const int am = argmax[e];
    for (int t = 0; t < nw; ++t) {
        const size_t r = R.row(v, t);
        dyh[r * C + c] = g[r * 2 * C + c] + (t == am ? s : 0.f);
    }
}

__global__ __launch_bounds__(256) void bn_segmax(const float *__restrict__ y, const float *__restrict__ mi,
                                                 float *__restrict__ out, int *__restrict__ argmax, int V, int C, Rows R,
                                                 FrameMap fm) {
    const long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (e >= (long long)V * C) return;
    const int v = (int)(e / C), c = (int)(e % C);
    mi += (size_t)fm_frame_of(fm, v) * 2 * C;
    const float m = mi[c], iv = mi[C + c];
    const int n = R.count(v) + (R.has_pad(v) ? 1 : 0);
    float best = -INFINITY;
    int bi = 0;
    for (int t = 0; t < n; ++t) {
        const float val = (y[R.row(v, t) * C + c] - m) * iv;
        if (val > best) { best = val; bi = t; }
    }
    out[e] = best;
    argmax[e] = bi;
}

__global__ __launch_bounds__(256) void segmax_bwd(const float *__restrict__ dfeat, const int *__restrict__ argmax,
                                                  float *__restrict__ dyh, int V, int C, Rows R) {
    const long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (e >= (long long)V * C) return;
    const int v = (int)(e / C), c = (int)(e % C);
    const float gval = dfeat[e];
    const int am = argmax[e];
    const int nw = R.vcnt ? R.count(v) + 1 : R.T;
    for (int t = 0; t < nw; ++t) dyh[R.row(v, t) * C + c] = (t == am) ? gval : 0.f;
}


// ---- float4 forms (channels a multiple of 4, 16-byte aligned tensors): one thread owns one (voxel, channel QUAD) and keeps FOUR
// rows of its voxel in flight per trip.  The scalar kernels above move 4 bytes per lane with one dependent row at a time and
// ran at 0.9-1.9 TB/s (config 2: 1.0 of 2.5 ms per step in these four kernels); the arithmetic and its order per channel -- first
// maximum wins, gradient sums in row order -- are unchanged, so results are bit-equal to the scalar forms.
constexpr int VU = 4;       // rows in flight per thread

__device__ __forceinline__ float4 ld4(const float *p) { return *(const float4 *)p; }
__device__ __forceinline__ void st4(float *p, const float4 v) { *(float4 *)p = v; }
__device__ __forceinline__ float4 bn4(const float4 y, const float4 m, const float4 iv) {
    return make_float4((y.x - m.x) * iv.x, (y.y - m.y) * iv.y, (y.z - m.z) * iv.z, (y.w - m.w) * iv.w);
}
#define MVX_ARGMAX4(val, t)                                                                                   \
    do {                                                                                                      \
        if (val.x > best.x) { best.x = val.x; bi.x = (t); }                                                   \
        if (val.y > best.y) { best.y = val.y; bi.y = (t); }                                                   \
        if (val.z > best.z) { best.z = val.z; bi.z = (t); }                                                   \
        if (val.w > best.w) { best.w = val.w; bi.w = (t); }                                                   \
    } while (0)

// rows of voxel v in compact or dense layout, resolved once per thread (voff / vcnt are the same for all lanes of a voxel)
struct VoxRows {
    size_t first, pad;      // first real row, the padded row (compact) -- dense: first row, unused
    int count, n, nw;       // real rows; rows that take part in the max; rows that are stored
    bool compact;
    __device__ __forceinline__ size_t row(int t) const { return (!compact || t < count) ? first + t : pad; }
};
__device__ __forceinline__ VoxRows vox_rows(const Rows &R, int v) {
    VoxRows w;
    w.compact = R.vcnt != nullptr;
    if (!w.compact) { w.first = (size_t)v * R.T; w.pad = 0; w.count = R.T; w.n = R.T; w.nw = R.T; return w; }
    w.count = R.vcnt[v];
    w.first = (size_t)R.voff[v];
    w.pad = (size_t)R.n_real + v;
    w.n = w.count + (w.count < R.T ? 1 : 0);
    w.nw = w.count + 1;                                   // the padded row is always written
    return w;
}

template <bool CONCAT>
__global__ __launch_bounds__(256) void vfe_bn_max4(const float *__restrict__ y, const float *__restrict__ mi,
                                                   float *__restrict__ out, int *__restrict__ argmax, int V, int C, Rows R,
                                                   FrameMap fm) {
    const int c4 = C >> 2;
    const long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (e >= (long long)V * c4) return;
    const int v = (int)(e / c4), c = (int)(e % c4) * 4;
    mi += (size_t)fm_frame_of(fm, v) * 2 * C;
    const float4 m = ld4(mi + c), iv = ld4(mi + C + c);
    const VoxRows w = vox_rows(R, v);
    float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    int4 bi = make_int4(0, 0, 0, 0);
    for (int t0 = 0; t0 < w.n; t0 += VU) {
        float4 q[VU];
#pragma unroll
        for (int u = 0; u < VU; ++u) q[u] = ld4(y + w.row(t0 + u < w.n ? t0 + u : w.n - 1) * C + c);
#pragma unroll
        for (int u = 0; u < VU; ++u)
            if (t0 + u < w.n) { const float4 val = bn4(q[u], m, iv); MVX_ARGMAX4(val, t0 + u); }
    }
    if (CONCAT) {
        for (int t0 = 0; t0 < w.nw; t0 += VU) {
            float4 q[VU];
#pragma unroll
            for (int u = 0; u < VU; ++u) q[u] = ld4(y + w.row(t0 + u < w.nw ? t0 + u : w.nw - 1) * C + c);
#pragma unroll
            for (int u = 0; u < VU; ++u)
                if (t0 + u < w.nw) {
                    float *o = out + w.row(t0 + u) * 2 * C;
                    st4(o + c, bn4(q[u], m, iv));
                    st4(o + C + c, best);
                }
        }
    } else {
        st4(out + (size_t)v * C + c, best);
    }
    *(int4 *)(argmax + (size_t)v * C + c) = bi;
}

__global__ __launch_bounds__(256) void vfe_max_concat_bwd4(const float *__restrict__ g, const int *__restrict__ argmax,
                                                           float *__restrict__ dyh, int V, int C, Rows R) {
    const int c4 = C >> 2;
    const long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (e >= (long long)V * c4) return;
    const int v = (int)(e / c4), c = (int)(e % c4) * 4;
    const VoxRows w = vox_rows(R, v);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t0 = 0; t0 < w.nw; t0 += VU) {                // the gradient of the max: summed in row order
        float4 q[VU];
#pragma unroll
        for (int u = 0; u < VU; ++u) q[u] = ld4(g + w.row(t0 + u < w.nw ? t0 + u : w.nw - 1) * 2 * C + C + c);
#pragma unroll
        for (int u = 0; u < VU; ++u)
            if (t0 + u < w.nw) { s.x += q[u].x; s.y += q[u].y; s.z += q[u].z; s.w += q[u].w; }
    }
    const int4 am = *(const int4 *)(argmax + (size_t)v * C + c);
    for (int t0 = 0; t0 < w.nw; t0 += VU) {
        float4 q[VU];
#pragma unroll
        for (int u = 0; u < VU; ++u) q[u] = ld4(g + w.row(t0 + u < w.nw ? t0 + u : w.nw - 1) * 2 * C + c);
#pragma unroll
        for (int u = 0; u < VU; ++u) {
            const int t = t0 + u;
            if (t < w.nw)
                st4(dyh + w.row(t) * C + c, make_float4(q[u].x + (t == am.x ? s.x : 0.f), q[u].y + (t == am.y ? s.y : 0.f),
                                                         q[u].z + (t == am.z ? s.z : 0.f), q[u].w + (t == am.w ? s.w : 0.f)));
        }
    }
}

__global__ __launch_bounds__(256) void segmax_bwd4(const float *__restrict__ dfeat, const int *__restrict__ argmax,
                                                   float *__restrict__ dyh, int V, int C, Rows R) {
    const int c4 = C >> 2;
    const long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (e >= (long long)V * c4) return;
    const int v = (int)(e / c4), c = (int)(e % c4) * 4;
    const VoxRows w = vox_rows(R, v);
    const float4 gv = ld4(dfeat + (size_t)v * C + c);
    const int4 am = *(const int4 *)(argmax + (size_t)v * C + c);
    for (int t = 0; t < w.nw; ++t)
        st4(dyh + w.row(t) * C + c, make_float4(t == am.x ? gv.x : 0.f, t == am.y ? gv.y : 0.f, t == am.z ? gv.z : 0.f,
                                                 t == am.w ? gv.w : 0.f));
}

// ---- compact row bookkeeping --------------------------------------------------------------------
// row_map [V*T] (dense row -> compact real row or -1)  ->  voff[v] = first compact row of voxel v,
// vcnt[v] = number of real rows, row_w[n_real + v] = T - vcnt[v] (weight of the padded row),
// row_w[real rows] = 1.
__global__ void voxel_row_offsets(const int *__restrict__ row_map, int V, int T, int n_real, int *__restrict__ voff,
                                  int *__restrict__ vcnt, float *__restrict__ row_w, float *__restrict__ fusion_row_w,
                                  FrameMap fm) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    // fusion layout (optional): real rows weigh 1, the shared padded row of frame f stands for all padded rows of f
    if (fusion_row_w) {
        for (int j = v; j < n_real; j += gridDim.x * blockDim.x) fusion_row_w[j] = 1.f;
        if (v < fm.F) {
            int nreal_f = 0;                          // real rows of frame f = extent of its real-row segment
            for (int sg = 0; sg < fm.nseg; ++sg)
                if (fm.seg_frame[sg] == v && fm.bound[sg + 1] <= n_real) nreal_f += fm.bound[sg + 1] - fm.bound[sg];
            fusion_row_w[n_real + v] = (float)(fm.count[v] - (double)nreal_f);
        }
    }
    if (v >= V) return;
    int first = -1, n = 0;
    for (int t = 0; t < T; ++t) {
        const int j = row_map[(size_t)v * T + t];
        if (j >= 0) { if (first < 0) first = j; row_w[j] = 1.f; ++n; }
    }
    voff[v] = first < 0 ? 0 : first;
    vcnt[v] = n;
    row_w[n_real + v] = (float)(T - n);
}

// VFE-1 input in compact form: real row j = [voxels[r][0:7], imfeat[j][0:F]], padded row of voxel v =
// [0 x 7, imfeat[n_real][0:F]] (imfeat's last row is the fusion output of the shared padded row).
// ld >= 7 + F: the row pitch of `out`; columns [7 + F, ld) are written as zeros (a pitch that is a multiple of 4 lets the row GEMM
// and its weight gradient read the rows with 16-byte loads: 23 floats per row made both fall to their scalar-load forms)
__global__ void vfe_compact_input(const float *__restrict__ vox, int vc, const int *__restrict__ rows_sel,
                                  const float *__restrict__ imfeat, int F, int n_real, int V, float *__restrict__ out,
                                  FrameMap fm, int ld) {
    const int W = 7 + F;
    const long long total = (long long)(n_real + V) * ld;
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(e / ld), c = (int)(e % ld);
        float val;
        if (c >= W) val = 0.f;
        else if (j < n_real) val = c < 7 ? vox[(size_t)rows_sel[j] * vc + c] : imfeat[(size_t)j * F + c - 7];
        else val = c < 7 ? 0.f : imfeat[(size_t)(n_real + fm_frame_of(fm, j - n_real)) * F + c - 7];   // its frame's shared padded row
        out[e] = val;
    }
}

// gradient wrt imfeat: real rows copy columns 7.., the shared padded row sums them over the voxels
__global__ __launch_bounds__(256) void vfe_compact_input_bwd(const float *__restrict__ g, int F, int n_real, int V,
                                                             float *__restrict__ dimfeat, double *__restrict__ padsum,
                                                             FrameMap fm) {
    const int W = 7 + F;
    const long long total = (long long)n_real * F;
    // four independent elements in flight per thread (one load per trip left the 4-byte gather latency-bound)
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e0 = blockIdx.x * (long long)blockDim.x + threadIdx.x; e0 < total; e0 += 4 * stride) {
        float q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long e = e0 + u * stride;
            const long long j = (e < total ? e : e0) / F;
            q[u] = g[(size_t)j * W + 7 + (int)((e < total ? e : e0) - j * F)];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (e0 + u * stride < total) dimfeat[e0 + u * stride] = q[u];
    }
    // padded rows: the padded rows of frame f feed the shared padded row of f.  ONE pass over the voxels with an LDS table of
    // per-(frame, column) f64 cells that goes out with one f64 atomic per touched cell.  (The first form made one pass PER FRAME, each with two barriers and a serial fold: 121 us for
    // 16 frames of 37 MB.)
    constexpr int MAXF = 64;                                  // columns the table holds (the fusion branch has 16)
    __shared__ double acc[MVX_MAX_FRAMES * MAXF];
    if (F <= MAXF && 256 % F == 0) {
        for (int i = threadIdx.x; i < fm.F * F; i += blockDim.x) acc[i] = 0.0;
        __syncthreads();
        // a workgroup takes a CONTIGUOUS run of voxels (one or two frames), thread (rt, ct) every rpi-th voxel of it at column ct:
        // a running f64 sum that goes to the table when the frame changes and at the end (<= 2-3 LDS atomics per thread)
        const int rpi = 256 / F, ct = threadIdx.x % F, rt = threadIdx.x / F;
        const long long per = ((long long)V + gridDim.x - 1) / gridDim.x;
        const long long lo = blockIdx.x * per, hi = lo + per < V ? lo + per : V;
        double run = 0.0;
        int cur = -1;
        for (long long v = lo + rt; v < hi; v += rpi) {
            const int f = fm_frame_of(fm, v);
            if (f != cur) {
                if (cur >= 0) atomicAdd(&acc[cur * F + ct], run);
                cur = f; run = 0.0;
            }
            run += (double)g[(size_t)(n_real + v) * W + 7 + ct];
        }
        if (cur >= 0) atomicAdd(&acc[cur * F + ct], run);
        __syncthreads();
        for (int i = threadIdx.x; i < fm.F * F; i += blockDim.x)
            if (acc[i] != 0.0) atomicAdd(padsum + i, acc[i]);
        return;
    }
    __shared__ float red[256];
    const int rpi = 256 / F;
    const int ct = threadIdx.x % F, rt = threadIdx.x / F;
    for (int sg = 0; sg < fm.nseg; ++sg) {
        const long long lo = fm.F == 1 ? 0 : fm.bound[sg], hi = fm.F == 1 ? V : fm.bound[sg + 1];
        const int f = fm.F == 1 ? 0 : (int)fm.seg_frame[sg];
        float s = 0.f;
        if (rt < rpi)
            for (long long v = lo + blockIdx.x * (long long)rpi + rt; v < hi; v += (long long)gridDim.x * rpi)
                s += g[(size_t)(n_real + v) * W + 7 + ct];
        __syncthreads();
        red[threadIdx.x] = s;
        __syncthreads();
        if (rt == 0) {
            double t = 0.0;
            for (int k = 0; k < rpi; ++k) t += (double)red[k * F + ct];
            atomicAdd(padsum + (size_t)f * F + ct, t);
        }
    }
}

__global__ void vfe_compact_pad_finish(const double *__restrict__ padsum, float *__restrict__ dimfeat, int n_real, int F,
                                       int n_frames) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < F * n_frames) dimfeat[(size_t)n_real * F + e] = (float)padsum[e];
}

}  // namespace

#define VFE_ARGS_OK (n_voxels >= 0 && t > 0 && channels > 0)
#define VFE_GRID dim3(mvx_cdiv((long long)n_voxels * channels, 256)), dim3(256), 0, (hipStream_t)stream

#define VFE_ROWS Rows{voff, vcnt, t, n_real}
#define VFE_GRID4 dim3(mvx_cdiv((long long)n_voxels * (channels / 4), 256)), dim3(256), 0, (hipStream_t)stream
static inline bool vfe_vec4(int channels, const void *a, const void *b, const void *c) {
    return channels % 4 == 0 && (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 15) == 0;
}
#define VFE_ROWS_OK ((voff == nullptr) == (vcnt == nullptr))
#define VFE_FRAMES(fm) FrameMap fm; MVX_CHECK_ARG(mvx_build_frame_map(fm, frames_host, frames_host ? MVX_ROWS_VOXELS : MVX_ROWS_SINGLE, n_voxels, 1.0))

extern "C" int mvx_vfe_bn_max_concat_frames(const float *y, const float *mean_inv, float *out, int32_t *argmax,
                                            int32_t n_voxels, int32_t t, int32_t channels, const int32_t *voff,
                                            const int32_t *vcnt, int32_t n_real, const mvx_frames_t *frames_host,
                                            void *stream) {
    MVX_CHECK_ARG(y && mean_inv && out && argmax && VFE_ARGS_OK && VFE_ROWS_OK);
    if (n_voxels == 0) return MVX_OK;
    VFE_FRAMES(fm);
    if (vfe_vec4(channels, y, out, argmax) && ((uintptr_t)mean_inv & 15) == 0)
        hipLaunchKernelGGL(vfe_bn_max4<true>, VFE_GRID4, y, mean_inv, out, argmax, n_voxels, channels, VFE_ROWS, fm);
    else
        hipLaunchKernelGGL(vfe_bn_max_concat, VFE_GRID, y, mean_inv, out, argmax, n_voxels, channels, VFE_ROWS, fm);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" int mvx_vfe_bn_max_concat(const float *y, const float *mean_inv, float *out, int32_t *argmax,
                                     int32_t n_voxels, int32_t t, int32_t channels, const int32_t *voff,
                                     const int32_t *vcnt, int32_t n_real, void *stream) {
    return mvx_vfe_bn_max_concat_frames(y, mean_inv, out, argmax, n_voxels, t, channels, voff, vcnt, n_real, nullptr, stream);
}

extern "C" int mvx_vfe_max_concat_backward(const float *grad_out, const int32_t *argmax, float *dyhat,
                                           int32_t n_voxels, int32_t t, int32_t channels, const int32_t *voff,
                                           const int32_t *vcnt, int32_t n_real, void *stream) {
    MVX_CHECK_ARG(grad_out && argmax && dyhat && VFE_ARGS_OK && VFE_ROWS_OK);
    if (n_voxels == 0) return MVX_OK;
    if (vfe_vec4(channels, grad_out, dyhat, argmax))
        hipLaunchKernelGGL(vfe_max_concat_bwd4, VFE_GRID4, grad_out, argmax, dyhat, n_voxels, channels, VFE_ROWS);
    else
        hipLaunchKernelGGL(vfe_max_concat_bwd, VFE_GRID, grad_out, argmax, dyhat, n_voxels, channels, VFE_ROWS);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" int mvx_bn_segment_max_frames(const float *y, const float *mean_inv, float *out, int32_t *argmax,
                                         int32_t n_voxels, int32_t t, int32_t channels, const int32_t *voff,
                                         const int32_t *vcnt, int32_t n_real, const mvx_frames_t *frames_host,
                                         void *stream) {
    MVX_CHECK_ARG(y && mean_inv && out && argmax && VFE_ARGS_OK && VFE_ROWS_OK);
    if (n_voxels == 0) return MVX_OK;
    VFE_FRAMES(fm);
    if (vfe_vec4(channels, y, out, argmax) && ((uintptr_t)mean_inv & 15) == 0)
        hipLaunchKernelGGL(vfe_bn_max4<false>, VFE_GRID4, y, mean_inv, out, argmax, n_voxels, channels, VFE_ROWS, fm);
    else
        hipLaunchKernelGGL(bn_segmax, VFE_GRID, y, mean_inv, out, argmax, n_voxels, channels, VFE_ROWS, fm);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" int mvx_bn_segment_max(const float *y, const float *mean_inv, float *out, int32_t *argmax,
                                  int32_t n_voxels, int32_t t, int32_t channels, const int32_t *voff,
                                  const int32_t *vcnt, int32_t n_real, void *stream) {
    return mvx_bn_segment_max_frames(y, mean_inv, out, argmax, n_voxels, t, channels, voff, vcnt, n_real, nullptr, stream);
}

extern "C" int mvx_segment_max_backward(const float *dfeat, const int32_t *argmax, float *dyhat, int32_t n_voxels,
                                        int32_t t, int32_t channels, const int32_t *voff, const int32_t *vcnt,
                                        int32_t n_real, void *stream) {
    MVX_CHECK_ARG(dfeat && argmax && dyhat && VFE_ARGS_OK && VFE_ROWS_OK);
    if (n_voxels == 0) return MVX_OK;
    if (vfe_vec4(channels, dfeat, dyhat, argmax))
        hipLaunchKernelGGL(segmax_bwd4, VFE_GRID4, dfeat, argmax, dyhat, n_voxels, channels, VFE_ROWS);
    else
        hipLaunchKernelGGL(segmax_bwd, VFE_GRID, dfeat, argmax, dyhat, n_voxels, channels, VFE_ROWS);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" int mvx_voxel_row_offsets_frames(const int32_t *row_map, int32_t n_voxels, int32_t t, int32_t n_real,
                                            int32_t *voff, int32_t *vcnt, float *row_w, float *fusion_row_w,
                                            const mvx_frames_t *frames_host, void *stream) {
    MVX_CHECK_ARG(row_map && voff && vcnt && row_w && n_voxels >= 0 && t > 0 && n_real >= 0);
    if (n_voxels == 0) return MVX_OK;
    FrameMap fm;
    // the FUSION layout's segment table carries the real-row extents the shared-row weights need
    MVX_CHECK_ARG(mvx_build_frame_map(fm, frames_host, frames_host ? MVX_ROWS_FUSION : MVX_ROWS_SINGLE,
                                      frames_host ? (long long)n_real + frames_host->n_frames : (long long)n_real + 1,
                                      (double)n_voxels * t));
    MVX_CHECK_ARG(!fusion_row_w || frames_host);
    hipLaunchKernelGGL(voxel_row_offsets, dim3(mvx_cdiv(n_voxels, 256)), dim3(256), 0, (hipStream_t)stream, row_map,
                       n_voxels, t, n_real, voff, vcnt, row_w, fusion_row_w, fm);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" int mvx_voxel_row_offsets(const int32_t *row_map, int32_t n_voxels, int32_t t, int32_t n_real,
                                     int32_t *voff, int32_t *vcnt, float *row_w, void *stream) {
    return mvx_voxel_row_offsets_frames(row_map, n_voxels, t, n_real, voff, vcnt, row_w, nullptr, nullptr, stream);
}

// ... with a row pitch: out is [n_real + n_voxels][ld], ld >= 7 + feat_channels, the columns beyond 7 + feat_channels zero
extern "C" int mvx_vfe_compact_input_pitch_frames(const float *voxels, int32_t vox_channels, const int32_t *rows_sel,
                                                  const float *imfeat, int32_t feat_channels, int32_t n_real, int32_t n_voxels,
                                                  float *out, int32_t ld, const mvx_frames_t *frames_host, void *stream) {
    MVX_CHECK_ARG(voxels && rows_sel && imfeat && out && vox_channels >= 7 && feat_channels > 0);
    MVX_CHECK_ARG(n_real >= 0 && n_voxels >= 0 && ld >= 7 + feat_channels);
    const long long total = (long long)(n_real + n_voxels) * ld;
    if (total == 0) return MVX_OK;
    VFE_FRAMES(fm);
    hipLaunchKernelGGL(vfe_compact_input, dim3(mvx_cdiv(total, 256) > 2048 ? 2048 : mvx_cdiv(total, 256)), dim3(256), 0,
                       (hipStream_t)stream, voxels, vox_channels, rows_sel, imfeat, feat_channels, n_real, n_voxels, out, fm, ld);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" int mvx_vfe_compact_input_frames(const float *voxels, int32_t vox_channels, const int32_t *rows_sel,
                                            const float *imfeat, int32_t feat_channels, int32_t n_real, int32_t n_voxels,
                                            float *out, const mvx_frames_t *frames_host, void *stream) {
    return mvx_vfe_compact_input_pitch_frames(voxels, vox_channels, rows_sel, imfeat, feat_channels, n_real, n_voxels, out,
                                              7 + feat_channels, frames_host, stream);
}

extern "C" int mvx_vfe_compact_input(const float *voxels, int32_t vox_channels, const int32_t *rows_sel,
                                     const float *imfeat, int32_t feat_channels, int32_t n_real, int32_t n_voxels,
                                     float *out, void *stream) {
    return mvx_vfe_compact_input_frames(voxels, vox_channels, rows_sel, imfeat, feat_channels, n_real, n_voxels, out, nullptr,
                                        stream);
}

extern "C" int mvx_vfe_compact_input_backward_frames(const float *grad_out, int32_t feat_channels, int32_t n_real,
                                                     int32_t n_voxels, float *dimfeat, double *scratch,
                                                     const mvx_frames_t *frames_host, void *stream) {
    MVX_CHECK_ARG(grad_out && dimfeat && scratch && feat_channels > 0 && feat_channels <= 256);
    MVX_CHECK_ARG(n_real >= 0 && n_voxels >= 0);
    VFE_FRAMES(fm);
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(scratch, 0, sizeof(double) * feat_channels * fm.F, st);
    if (e != hipSuccess) return (int)e;
    const long long cells = (long long)n_real * feat_channels;
    const unsigned grid = (unsigned)(mvx_cdiv(cells, 1024) > 2048 ? 2048 : (mvx_cdiv(cells, 1024) < 256 ? 256 : mvx_cdiv(cells, 1024)));
    hipLaunchKernelGGL(vfe_compact_input_bwd, dim3(grid), dim3(256), 0, st, grad_out, feat_channels, n_real, n_voxels,
                       dimfeat, scratch, fm);
    MVX_LAUNCH_CHECK();
    hipLaunchKernelGGL(vfe_compact_pad_finish, dim3(mvx_cdiv(feat_channels * fm.F, 64)), dim3(64), 0, st,
                       (const double *)scratch, dimfeat, n_real, feat_channels, fm.F);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" int mvx_vfe_compact_input_backward(const float *grad_out, int32_t feat_channels, int32_t n_real,
                                              int32_t n_voxels, float *dimfeat, double *scratch, void *stream) {
    return mvx_vfe_compact_input_backward_frames(grad_out, feat_channels, n_real, n_voxels, dimfeat, scratch, nullptr, stream);
}
