// Point <-> image fusion: per-point 4-tap sampling of the FPN maps at the projected pixel.
//
// Replaces featureMaping (modules/imhead/Pipe.py:23-82), forward only (no gradient flows to the
// frozen extractor or to the projections).  Arithmetic as written there, in f32:
//     q = proj / (imsize / feat_hw) - eps ;  i = trunc(q) ;  f = q - i
//     out = F[i,j]*fx*fy + F[i+1,j]*(1-fx)*fy + F[i,j+1]*fx*(1-fy) + F[i+1,j+1]*(1-fx)*(1-fy)
// (the weights are the reference's, "inverted" relative to textbook bilinear, Pipe.py:72-75);
// rows whose x == y == z == 0 are padding: their 9 voxel channels are zeroed IN PLACE (Pipe.py:54-59)
// and their output is 0 (Pipe.py:80).  The maps are zero-padded by one row/column (Pipe.py:47-48).
//
// Feature maps are channels-last [H][W][C] so every tap is one contiguous C*4-byte run: a wave
// reads it as 64 lanes x float4 (1 KiB per instruction at C = 256) and writes the output row
// segment the same way.
//
// Compact mode: the padded rows of a frame are all identical (zeros), so only the real rows are
// sampled; row_map sends a dense row to its compact row (or -1), and one shared zero row stands
// for every padded row downstream.
#include "common.h"
#include "split_common.h"

namespace {

constexpr int MAX_LEVELS = 4;
struct Levels {
    const float *feat[MAX_LEVELS];
    int h[MAX_LEVELS], w[MAX_LEVELS];
    int n;
};
// the same levels for every frame of a frame set (maps of frame f: feat[f * n + level]; one shape per level)
struct FrameLevels {
    const float *feat[MVX_MAX_FRAMES * MAX_LEVELS];
    int h[MAX_LEVELS], w[MAX_LEVELS];
    int n;
};

__device__ __forceinline__ bool row_is_zero(const float *v) { return v[0] == 0.f && v[1] == 0.f && v[2] == 0.f; }

// ---- dense-row -> compact-row map (three phases: block counts, scan of counts, map) -----------
__global__ __launch_bounds__(256) void map_count(const float *__restrict__ vox, int vc, long long R, int *__restrict__ bcount) {
    __shared__ int s[4];
    const long long r = blockIdx.x * 256ll + threadIdx.x;
    const int flag = (r < R) && !row_is_zero(vox + r * vc);
    const unsigned long long b = __ballot(flag);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = __popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) bcount[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

__global__ __launch_bounds__(1024) void map_scan(int *__restrict__ bcount, int nblocks, int *__restrict__ n_real) {
    __shared__ int smem[17];
    int base = 0;
    for (int t0 = 0; t0 < nblocks; t0 += 1024) {
        const int i = t0 + threadIdx.x;
        const int v = i < nblocks ? bcount[i] : 0;
        int tot;
        const int ex = block_excl_scan_i32(v, smem, &tot);
        if (i < nblocks) bcount[i] = base + ex;
        base += tot;
    }
    if (threadIdx.x == 0) *n_real = base;
}

__global__ __launch_bounds__(256) void map_write(float *__restrict__ vox, int vc, long long R,
                                                 const int *__restrict__ boff, int *__restrict__ row_map,
                                                 int *__restrict__ rows_sel, int zero_padding) {
    __shared__ int s[4];
    const long long r = blockIdx.x * 256ll + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int flag = (r < R) && !row_is_zero(vox + r * vc);
    const unsigned long long b = __ballot(flag);
    if (lane == 0) s[wv] = __popcll(b);
    __syncthreads();
    int off = boff[blockIdx.x];
    for (int k = 0; k < wv; ++k) off += s[k];
    off += __popcll(b & ((1ull << lane) - 1ull));
    if (r < R) {
        row_map[r] = flag ? off : -1;
        if (flag && rows_sel) rows_sel[off] = (int)r;
        // padding rows: the reference zeroes all their channels in place (Pipe.py:54-59); done here, where every
        // row is visited anyway, so that the compact sampler only has to touch the real rows
        if (!flag && zero_padding)
            for (int c = 3; c < vc; ++c) vox[r * vc + c] = 0.f;
    }
}

// real_off[f] = rank of the first real row at or after the first dense row of frame f (= real rows before frame f)
__global__ void map_frame_offsets(const int *__restrict__ row_map, long long R, const int *__restrict__ n_real, int T,
                                  FrameMap fm, int *__restrict__ real_off) {
    const int f = threadIdx.x;
    if (f > fm.F) return;
    if (f == fm.F) { real_off[f] = *n_real; return; }
    long long r = (long long)fm.bound[f] * T;       // fm: MVX_ROWS_VOXELS segments (first voxel of frame f)
    while (r < R && row_map[r] < 0) ++r;
    real_off[f] = r < R ? row_map[r] : *n_real;
}

// ---- sampling: one wave per (dense row, level) ---------------------------------------------------
__global__ __launch_bounds__(256) void feature_sample(float *__restrict__ vox, int vc, long long R,
                                                      const int *__restrict__ row_map, Levels L, int C,
                                                      float im_h, float im_w, float eps, float *__restrict__ out,
                                                      int *__restrict__ status) {
    const long long wave = (blockIdx.x * 256ll + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const long long r = wave / L.n;
    const int lv = (int)(wave % L.n);
    if (r >= R) return;
    float *v = vox + r * vc;
    const bool zero = row_is_zero(v);
    const long long orow = row_map ? row_map[r] : r;
    const int ldo = L.n * C;
    if (zero) {
        // all waves of this row see the same (still unmodified) xyz only if the zeroing happens
        // after every wave tested it: level 0's wave zeroes the projection columns and the
        // payload columns 3.. ; xyz are zero already.
        if (lv == 0 && lane >= 3 && lane < vc) v[lane] = 0.f;
        if (orow >= 0 && !row_map)
            for (int c = lane * 4; c < C; c += 256) *(float4 *)(out + orow * ldo + lv * C + c) = make_float4(0, 0, 0, 0);
        return;
    }
    if (orow < 0) return;
    const int H = L.h[lv], W = L.w[lv];
    // regionSize = imsize / feat_hw (f32), q = proj / regionSize - eps  (Pipe.py:44,62)
    const float qy = v[vc - 2] / (im_h / (float)H) - eps;
    const float qx = v[vc - 1] / (im_w / (float)W) - eps;
    const long long iy = (long long)qy, ix = (long long)qx;      // .long(): truncation toward zero
    const float fy = qy - (float)iy, fx = qx - (float)ix;        // Pipe.py:64-65 (first index = row)
    // the reference indexes the padded map and asserts i+1 < H+1 (Pipe.py:71); negative indices
    // would wrap in torch -- flag both instead of reading out of bounds
    if (iy < 0 || ix < 0 || iy + 1 > H || ix + 1 > W) {
        if (lane == 0) atomicOr(status, 1);
        for (int c = lane * 4; c < C; c += 256) *(float4 *)(out + orow * ldo + lv * C + c) = make_float4(0, 0, 0, 0);
        return;
    }
    const float *F = L.feat[lv];
    const bool y0 = iy < H, y1 = iy + 1 < H, x0 = ix < W, x1 = ix + 1 < W;
    const float xi = fy, yi = fx;              // reference names: xi <- first (row) coordinate
    const float xi_ = 1.f - xi, yi_ = 1.f - yi;
    for (int c = lane * 4; c < C; c += 256) {
        const float4 z = make_float4(0, 0, 0, 0);
        const float4 f00 = (y0 && x0) ? *(const float4 *)(F + ((size_t)iy * W + ix) * C + c) : z;
        const float4 f10 = (y1 && x0) ? *(const float4 *)(F + ((size_t)(iy + 1) * W + ix) * C + c) : z;
        const float4 f01 = (y0 && x1) ? *(const float4 *)(F + ((size_t)iy * W + ix + 1) * C + c) : z;
        const float4 f11 = (y1 && x1) ? *(const float4 *)(F + ((size_t)(iy + 1) * W + ix + 1) * C + c) : z;
        float4 o;
#define MVX_TAP(m) o.m = (((f00.m * xi) * yi + (f10.m * xi_) * yi) + (f01.m * xi) * yi_) + (f11.m * xi_) * yi_;
        MVX_TAP(x) MVX_TAP(y) MVX_TAP(z) MVX_TAP(w)
#undef MVX_TAP
        *(float4 *)(out + orow * ldo + lv * C + c) = o;
    }
}

// Compact form: one wave per (REAL row j, level); rows_sel[j] is the dense row, the output row is j.  The padding
// rows are not visited at all (88 % of the dense rows on a lidar frame); same arithmetic as feature_sample.
__global__ __launch_bounds__(256) void feature_sample_rows(const float *__restrict__ vox, int vc, const int *__restrict__ rows_sel,
                                                           int n_real, FrameLevels L, int C, float im_h, float im_w, float eps,
                                                           float *__restrict__ out, int *__restrict__ status, FrameMap fm,
                                                           unsigned *__restrict__ amax_slot, unsigned short *__restrict__ planes,
                                                           long long plane_rows) {
    // planes (may be NULL): u16 [3][plane_rows][levels * C], the rows ALSO written as their three bf16 pieces (hi + mid + lo = the
    // f32 value exactly): the operand format of the first fusion layer's weight gradient (rowgemm_pre.hip) formed while the rows
    // are in registers instead of by a pass of mvx_split_rows over them (245 MB read back per 4-frame step)
    // amax_slot (may be NULL): raised to max |sampled value| -- the range tag of the rows for the fp16x3 arithmetic, formed while
    // they are written instead of by a pass of mvx_tensor_amax over them (108 us per 245 MB beside the step's other kernels)
    const long long wave = (blockIdx.x * 256ll + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const long long j = wave / L.n;
    const int lv = (int)(wave % L.n);
    if (j >= n_real) return;
    const float *v = vox + (size_t)rows_sel[j] * vc;
    const int ldo = L.n * C;
    const int H = L.h[lv], W = L.w[lv];
    const float qy = v[vc - 2] / (im_h / (float)H) - eps;
    const float qx = v[vc - 1] / (im_w / (float)W) - eps;
    const long long iy = (long long)qy, ix = (long long)qx;
    const float fy = qy - (float)iy, fx = qx - (float)ix;
    if (iy < 0 || ix < 0 || iy + 1 > H || ix + 1 > W) {
        if (lane == 0) atomicOr(status, 1);
        for (int c = lane * 4; c < C; c += 256) {
            if (out) *(float4 *)(out + j * ldo + lv * C + c) = make_float4(0, 0, 0, 0);
            if (planes)
                for (int q = 0; q < 3; ++q) *(uint2 *)(planes + ((size_t)q * plane_rows + j) * ldo + lv * C + c) = make_uint2(0u, 0u);
        }
        return;
    }
    const float *F = L.feat[fm_frame_of(fm, j) * L.n + lv];       // the maps of the row's own frame
    const bool y0 = iy < H, y1 = iy + 1 < H, x0 = ix < W, x1 = ix + 1 < W;
    const float xi = fy, yi = fx;
    const float xi_ = 1.f - xi, yi_ = 1.f - yi;
    float mx = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
        const float4 z = make_float4(0, 0, 0, 0);
        const float4 f00 = (y0 && x0) ? *(const float4 *)(F + ((size_t)iy * W + ix) * C + c) : z;
        const float4 f10 = (y1 && x0) ? *(const float4 *)(F + ((size_t)(iy + 1) * W + ix) * C + c) : z;
        const float4 f01 = (y0 && x1) ? *(const float4 *)(F + ((size_t)iy * W + ix + 1) * C + c) : z;
        const float4 f11 = (y1 && x1) ? *(const float4 *)(F + ((size_t)(iy + 1) * W + ix + 1) * C + c) : z;
        float4 o;
#define MVX_TAP(m) o.m = (((f00.m * xi) * yi + (f10.m * xi_) * yi) + (f01.m * xi) * yi_) + (f11.m * xi_) * yi_;
        MVX_TAP(x) MVX_TAP(y) MVX_TAP(z) MVX_TAP(w)
#undef MVX_TAP
        if (out) *(float4 *)(out + j * ldo + lv * C + c) = o;        // out == NULL: planes only (their reader needs no f32 rows)
        if (planes) {
            uint2 pc[3];
            split_n<3, 0>(o.x, o.y, o.z, o.w, pc);
#pragma unroll
            for (int q = 0; q < 3; ++q) *(uint2 *)(planes + ((size_t)q * plane_rows + j) * ldo + lv * C + c) = pc[q];
        }
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
    }
    if (amax_slot) mvx_wave_amax_to(amax_slot, mx);
}

// ---- compact rows -> dense rows and back ----------------------------------------------------------
__global__ void expand_rows(const float *__restrict__ compact, const int *__restrict__ row_map, int pad_row,
                            float *__restrict__ out, long long R, int C) {
    const long long total = R * C;
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long r = e / C;
        const int c = (int)(e % C);
        const int j = row_map[r];
        out[e] = compact[(size_t)(j >= 0 ? j : pad_row) * C + c];
    }
}

// real rows: copy; padded rows: column sums (block partials in f32, f64 atomics across blocks)
__global__ __launch_bounds__(256) void expand_rows_bwd(const float *__restrict__ g, const int *__restrict__ row_map,
                                                       float *__restrict__ dcompact, double *__restrict__ padsum,
                                                       long long R, int C) {
    __shared__ float red[256];
    const int rpi = 256 / C;                       // C <= 256
    const int ct = threadIdx.x % C, rt = threadIdx.x / C;
    float s = 0.f;
    if (rt < rpi) {
        for (long long r = blockIdx.x * (long long)rpi + rt; r < R; r += (long long)gridDim.x * rpi) {
            const float val = g[r * C + ct];
            const int j = row_map[r];
            if (j >= 0) dcompact[(size_t)j * C + ct] = val; else s += val;
        }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (rt == 0) {
        double t = 0.0;
        for (int k = 0; k < rpi; ++k) t += (double)red[k * C + ct];
        atomicAdd(padsum + ct, t);
    }
}

__global__ void pad_finish(const double *__restrict__ padsum, float *__restrict__ dcompact, int pad_row, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) dcompact[(size_t)pad_row * C + c] = (float)padsum[c];
}

}  // namespace

extern "C" size_t mvx_row_compact_workspace_bytes(int64_t rows) {
    return (size_t)(mvx_cdiv(rows > 0 ? rows : 1, 256) + 1) * sizeof(int32_t);
}

extern "C" int mvx_row_compact_map_frames(float *voxels, int32_t vox_channels, int64_t rows, int32_t *row_map,
                                          int32_t *rows_sel, int32_t *n_real, void *workspace, size_t workspace_bytes,
                                          const mvx_frames_t *frames_host, int32_t *real_off, void *stream) {
    MVX_CHECK_ARG(voxels && row_map && n_real && workspace && vox_channels >= 3 && rows >= 0);
    MVX_CHECK_ARG(workspace_bytes >= mvx_row_compact_workspace_bytes(rows));
    MVX_CHECK_ARG((frames_host == nullptr) == (real_off == nullptr));
    hipStream_t st = (hipStream_t)stream;
    FrameMap fm;
    if (frames_host) {
        MVX_CHECK_ARG(frames_host->t > 0 && rows % frames_host->t == 0);
        MVX_CHECK_ARG(mvx_build_frame_map(fm, frames_host, MVX_ROWS_VOXELS, rows / frames_host->t, 1.0));
    }
    if (rows == 0) {
        hipError_t e = hipMemsetAsync(n_real, 0, sizeof(int32_t), st);
        if (e == hipSuccess && real_off) e = hipMemsetAsync(real_off, 0, sizeof(int32_t) * (frames_host->n_frames + 1), st);
        return e == hipSuccess ? MVX_OK : (int)e;
    }
    const unsigned nb = mvx_cdiv(rows, 256);
    int *bc = (int *)workspace;
    hipLaunchKernelGGL(map_count, dim3(nb), dim3(256), 0, st, voxels, vox_channels, (long long)rows, bc);
    MVX_LAUNCH_CHECK();
    hipLaunchKernelGGL(map_scan, dim3(1), dim3(1024), 0, st, bc, (int)nb, n_real);
    MVX_LAUNCH_CHECK();
    hipLaunchKernelGGL(map_write, dim3(nb), dim3(256), 0, st, voxels, vox_channels, (long long)rows, (const int *)bc,
                       row_map, rows_sel, 1);
    MVX_LAUNCH_CHECK();
    if (real_off) {
        hipLaunchKernelGGL(map_frame_offsets, dim3(1), dim3(64), 0, st, (const int *)row_map, (long long)rows,
                           (const int *)n_real, frames_host->t, fm, real_off);
        MVX_LAUNCH_CHECK();
    }
    return MVX_OK;
}

extern "C" int mvx_row_compact_map(float *voxels, int32_t vox_channels, int64_t rows, int32_t *row_map,
                                   int32_t *rows_sel, int32_t *n_real, void *workspace, size_t workspace_bytes,
                                   void *stream) {
    return mvx_row_compact_map_frames(voxels, vox_channels, rows, row_map, rows_sel, n_real, workspace, workspace_bytes,
                                      nullptr, nullptr, stream);
}

extern "C" int mvx_feature_sample(float *voxels, int32_t vox_channels, int64_t rows, const int32_t *row_map,
                                  const float *const *feats_host, const int32_t *feat_hw_host, int32_t n_levels,
                                  int32_t channels, float imsize_h, float imsize_w, float eps, float *out,
                                  int32_t *status, void *stream) {
    MVX_CHECK_ARG(voxels && feats_host && feat_hw_host && out && status && rows >= 0);
    MVX_CHECK_ARG(vox_channels >= 5 && vox_channels <= 64 && n_levels >= 1 && n_levels <= MAX_LEVELS);
    MVX_CHECK_ARG(channels > 0 && channels % 4 == 0);
    if (rows == 0) return MVX_OK;
    Levels L;
    L.n = n_levels;
    for (int k = 0; k < MAX_LEVELS; ++k) {
        L.feat[k] = k < n_levels ? feats_host[k] : nullptr;
        L.h[k] = k < n_levels ? feat_hw_host[2 * k] : 0;
        L.w[k] = k < n_levels ? feat_hw_host[2 * k + 1] : 0;
        if (k < n_levels) MVX_CHECK_ARG(L.feat[k] && L.h[k] > 0 && L.w[k] > 0);
    }
    const long long waves = (long long)rows * n_levels;
    hipLaunchKernelGGL(feature_sample, dim3(mvx_cdiv(waves, 4)), dim3(256), 0, (hipStream_t)stream, voxels, vox_channels,
                       (long long)rows, row_map, L, channels, imsize_h, imsize_w, eps, out, status);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

static int feature_sample_rows_impl(const float *voxels, int32_t vox_channels, const int32_t *rows_sel,
                                    int32_t n_real, const float *const *feats_host, const int32_t *feat_hw_host,
                                    int32_t n_levels, int32_t channels, float imsize_h, float imsize_w, float eps,
                                    float *out, int32_t *status, const mvx_frames_t *frames_host, float *out_amax,
                                    void *planes, int64_t plane_rows, void *stream) {
    MVX_CHECK_ARG(voxels && rows_sel && feats_host && feat_hw_host && (out || planes) && status && n_real >= 0);
    MVX_CHECK_ARG(!planes || (plane_rows >= n_real && (((uintptr_t)planes) & 7) == 0));
    MVX_CHECK_ARG(vox_channels >= 5 && vox_channels <= 64 && n_levels >= 1 && n_levels <= MAX_LEVELS);
    MVX_CHECK_ARG(channels > 0 && channels % 4 == 0);
    if (n_real == 0) return MVX_OK;
    FrameMap fm;
    MVX_CHECK_ARG(mvx_build_frame_map(fm, frames_host, frames_host ? MVX_ROWS_REAL : MVX_ROWS_SINGLE, n_real, 1.0));
    FrameLevels L;
    L.n = n_levels;
    for (int k = 0; k < MAX_LEVELS; ++k) {
        L.h[k] = k < n_levels ? feat_hw_host[2 * k] : 0;
        L.w[k] = k < n_levels ? feat_hw_host[2 * k + 1] : 0;
        if (k < n_levels) MVX_CHECK_ARG(L.h[k] > 0 && L.w[k] > 0);
    }
    for (int k = 0; k < MVX_MAX_FRAMES * MAX_LEVELS; ++k) L.feat[k] = nullptr;
    for (int k = 0; k < fm.F * n_levels; ++k) {
        L.feat[k] = feats_host[k];                 // frame-major: feats_host[f * n_levels + level]
        MVX_CHECK_ARG(L.feat[k]);
    }
    const long long waves = (long long)n_real * n_levels;
    hipLaunchKernelGGL(feature_sample_rows, dim3(mvx_cdiv(waves, 4)), dim3(256), 0, (hipStream_t)stream, voxels, vox_channels,
                       rows_sel, n_real, L, channels, imsize_h, imsize_w, eps, out, status, fm, (unsigned *)out_amax,
                       (unsigned short *)planes, (long long)plane_rows);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" int mvx_feature_sample_rows_frames(const float *voxels, int32_t vox_channels, const int32_t *rows_sel,
                                              int32_t n_real, const float *const *feats_host, const int32_t *feat_hw_host,
                                              int32_t n_levels, int32_t channels, float imsize_h, float imsize_w, float eps,
                                              float *out, int32_t *status, const mvx_frames_t *frames_host, float *out_amax,
                                              void *stream) {
    return feature_sample_rows_impl(voxels, vox_channels, rows_sel, n_real, feats_host, feat_hw_host, n_levels, channels, imsize_h,
                                    imsize_w, eps, out, status, frames_host, out_amax, nullptr, 0, stream);
}

// ... which also writes the rows as planes of bf16 pieces, u16 [3][plane_rows][n_levels * channels] (rows [0, n_real); the caller
// clears the others): what mvx_split_rows would make of `out`, bit for bit, without reading it back
extern "C" int mvx_feature_sample_rows_planes_frames(const float *voxels, int32_t vox_channels, const int32_t *rows_sel,
                                                     int32_t n_real, const float *const *feats_host, const int32_t *feat_hw_host,
                                                     int32_t n_levels, int32_t channels, float imsize_h, float imsize_w, float eps,
                                                     float *out, int32_t *status, const mvx_frames_t *frames_host, float *out_amax,
                                                     void *planes, int64_t plane_rows, void *stream) {
    MVX_CHECK_ARG(planes);
    return feature_sample_rows_impl(voxels, vox_channels, rows_sel, n_real, feats_host, feat_hw_host, n_levels, channels, imsize_h,
                                    imsize_w, eps, out, status, frames_host, out_amax, planes, plane_rows, stream);
}

extern "C" int mvx_feature_sample_rows(const float *voxels, int32_t vox_channels, const int32_t *rows_sel, int32_t n_real,
                                       const float *const *feats_host, const int32_t *feat_hw_host, int32_t n_levels,
                                       int32_t channels, float imsize_h, float imsize_w, float eps, float *out,
                                       int32_t *status, void *stream) {
    return mvx_feature_sample_rows_frames(voxels, vox_channels, rows_sel, n_real, feats_host, feat_hw_host, n_levels, channels,
                                          imsize_h, imsize_w, eps, out, status, nullptr, nullptr, stream);
}

extern "C" int mvx_expand_rows(const float *compact, const int32_t *row_map, int32_t pad_row, float *out,
                               int64_t rows, int32_t channels, void *stream) {
    MVX_CHECK_ARG(compact && row_map && out && rows >= 0 && channels > 0 && pad_row >= 0);
    if (rows == 0) return MVX_OK;
    const long long total = (long long)rows * channels;
    hipLaunchKernelGGL(expand_rows, dim3(mvx_cdiv(total, 256) > 4096 ? 4096 : mvx_cdiv(total, 256)), dim3(256), 0,
                       (hipStream_t)stream, compact, row_map, pad_row, out, (long long)rows, channels);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" int mvx_expand_rows_backward(const float *grad_out, const int32_t *row_map, int32_t pad_row,
                                        float *dcompact, double *scratch, int64_t rows, int32_t channels,
                                        void *stream) {
    MVX_CHECK_ARG(grad_out && row_map && dcompact && scratch && rows >= 0 && pad_row >= 0);
    MVX_CHECK_ARG(channels > 0 && channels <= 256);
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(scratch, 0, sizeof(double) * channels, st);
    if (e != hipSuccess) return (int)e;
    if (rows > 0) {
        const int rpi = 256 / channels;
        const long long nb = (rows + rpi - 1) / rpi;
        hipLaunchKernelGGL(expand_rows_bwd, dim3((unsigned)(nb > 1024 ? 1024 : nb)), dim3(256), 0, st, grad_out, row_map,
                           dcompact, scratch, (long long)rows, channels);
        MVX_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(pad_finish, dim3(mvx_cdiv(channels, 64)), dim3(64), 0, st, (const double *)scratch, dcompact,
                       pad_row, channels);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}
