// Data-movement kernels of the region proposal network on frame sets (modules/voxelnet/Pipe.py:45-75,
// modules/layers/Blocks.py:31-51).  The arithmetic of the RPN runs on the matrix-core kernels of conv3d.hip (3x3 blocks)
// and linear.hip (transposed convolutions with stride = kernel, the two 1x1 heads); what is left is layout:
//
//   space_to_depth   a stride-2 3x3 convolution is a stride-1 convolution with a 2x2 window on the image whose pixel (y, x)
//                    holds the four input pixels (2y+pr, 2x+pc): channel block p = 2 pr + pc.  The input may be spread
//                    over `planes` depth planes per frame (the CML output [F*2][H][W][64] is the (1,128,H,W) BEV map of
//                    VoxelNet.py:36 with channel = c*2+d; here the planes are simply laid side by side, d-major, and the
//                    first RPN weight is permuted to match), so the BEV reshape costs no pass of its own.
//   depth_to_space   the inverse (gradient of the above).
//   d2s_bn_apply     ConvTranspose2d with kernel = stride = s is a row GEMM x[site][ci] -> t[site][(i,j)][co]; this kernel
//                    normalises t (BatchNorm of Blocks.py:50, per frame) and writes pixel (y*s+i, x*s+j) of the up-sampled
//                    map into a channel slice of the 768-channel concat buffer (Pipe.py:72) in one pass.
//   s2d_gather       the inverse read (gradient of the concat slice back into GEMM layout).
//   bn_apply_strided BatchNorm apply of a dense [rows][C] tensor into a channel slice of a wider buffer.
#include "common.h"

namespace {

// in [F*P][H][W][C] -> out [F][H/2][W/2][4][P][C]           (reverse: out -> in)
__global__ void space_to_depth(const float *__restrict__ src, float *__restrict__ dst, int F, int P, int H, int W, int C,
                               int reverse) {
    const int c4 = C >> 2, H2 = H >> 1, W2 = W >> 1;
    const size_t total = (size_t)F * H2 * W2 * 4 * P * c4;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        size_t r = e;
        const int c = (int)(r % c4); r /= c4;
        const int d = (int)(r % P); r /= P;
        const int p = (int)(r % 4); r /= 4;
        const int x = (int)(r % W2); r /= W2;
        const int y = (int)(r % H2);
        const int f = (int)(r / H2);
        const size_t full = ((((size_t)(f * P + d) * H + 2 * y + (p >> 1)) * W + 2 * x + (p & 1)) * C) + c * 4;
        if (!reverse) ((float4 *)dst)[e] = *(const float4 *)(src + full);
        else *(float4 *)(dst + full) = ((const float4 *)src)[e];
    }
}

// t [F][h][w][s*s][C] (GEMM layout) -> out [F][h*s][w*s] rows of ld_out floats, columns [col_off, col_off + C):
// out = (t - mean_f) * inv_f         (mi == nullptr: plain copy)
// reverse: t[...] = out slice        (gradient gather; mi unused)
__global__ void d2s_rows(float *__restrict__ t, const float *__restrict__ mi, float *__restrict__ out, int F, int h, int w,
                         int s, int C, int ld_out, int col_off, int reverse) {
    const int c4 = C >> 2;
    const size_t total = (size_t)F * h * w * s * s * c4;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        size_t r = e;
        const int c = (int)(r % c4) * 4; r /= c4;
        const int ij = (int)(r % (s * s)); r /= (s * s);
        const int x = (int)(r % w); r /= w;
        const int y = (int)(r % h);
        const int f = (int)(r / h);
        const int i = ij / s, j = ij - i * s;
        float *o = out + ((((size_t)f * h * s + (size_t)y * s + i) * ((size_t)w * s)) + (size_t)x * s + j) * ld_out + col_off + c;
        if (reverse) {
            ((float4 *)t)[e] = *(const float4 *)o;     // t is written in this direction
            continue;
        }
        float4 v = ((const float4 *)t)[e];
        if (mi) {
            const float *fmi = mi + (size_t)f * 2 * C;
            const float4 m = *(const float4 *)(fmi + c), iv = *(const float4 *)(fmi + C + c);
            v.x = (v.x - m.x) * iv.x; v.y = (v.y - m.y) * iv.y; v.z = (v.z - m.z) * iv.z; v.w = (v.w - m.w) * iv.w;
        }
        *(float4 *)o = v;
    }
}

// y [rows][C] dense -> out rows of ld_out floats, columns [col_off, col_off + C): (y - mean_f) * inv_f, frames = equal shares
// of the rows.  reverse: dense[rows][C] = out slice (no arithmetic).
__global__ void bn_apply_strided(float *__restrict__ y, const float *__restrict__ mi, float *__restrict__ out, size_t rows,
                                 size_t rows_per_frame, int C, int ld_out, int col_off, int reverse) {
    const int c4 = C >> 2;
    const size_t total = rows * c4;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t row = e / c4;
        const int c = (int)(e % c4) * 4;
        float *o = out + row * ld_out + col_off + c;
        if (reverse) {
            ((float4 *)y)[e] = *(const float4 *)o;
            continue;
        }
        float4 v = ((const float4 *)y)[e];
        if (mi) {
            const float *fmi = mi + (row / rows_per_frame) * 2 * C;
            const float4 m = *(const float4 *)(fmi + c), iv = *(const float4 *)(fmi + C + c);
            v.x = (v.x - m.x) * iv.x; v.y = (v.y - m.y) * iv.y; v.z = (v.z - m.z) * iv.z; v.w = (v.w - m.w) * iv.w;
        }
        *(float4 *)o = v;
    }
}

// per-channel (sum, sum of squares) of a [F][rows_per_frame][C] tensor, per frame (replicated accumulators)
__global__ __launch_bounds__(256) void row_stats_frames(const float *__restrict__ y, double *__restrict__ stats,
                                                        size_t rows_per_frame, int C) {
    __shared__ double red[2][256][4];
    const int c4 = C >> 2;
    const int rpi = max(1, 256 / c4);
    const int ct = threadIdx.x % c4, rt = threadIdx.x / c4;
    const int f = blockIdx.y;
    y += (size_t)f * rows_per_frame * C;
    stats += (size_t)f * MVX_REP * 2 * C;
    for (int cb = 0; cb < c4; cb += 256) {
        const int col = cb + ct;
        double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};      // f64 throughout (var = E[y^2] - mean^2 cancels)
        if (rt < rpi && col < c4) {
            for (size_t r = blockIdx.x * (size_t)rpi + rt; r < rows_per_frame; r += (size_t)gridDim.x * rpi) {
                const float4 v = *(const float4 *)(y + r * C + col * 4);
                const double d[4] = {(double)v.x, (double)v.y, (double)v.z, (double)v.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) { s1[j] += d[j]; s2[j] += d[j] * d[j]; }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { red[0][threadIdx.x][j] = s1[j]; red[1][threadIdx.x][j] = s2[j]; }
        __syncthreads();
        if (rt == 0 && col < c4) {
            for (int k = 0; k < 2; ++k)
                for (int j = 0; j < 4; ++j) {
                    double t = 0.0;
                    for (int r = 0; r < rpi; ++r) t += red[k][r * c4 + ct][j];
                    atomicAdd(stats + ((size_t)(blockIdx.x % MVX_REP) * 2 + k) * C + col * 4 + j, t);
                }
        }
        __syncthreads();
    }
}

inline unsigned ew_grid(size_t n) { return (unsigned)(mvx_cdiv(n, 256) > 8192 ? 8192 : mvx_cdiv(n, 256)); }

}  // namespace

extern "C" int mvx_space_to_depth_frames(const float *in, float *out, int32_t n_frames, int32_t planes, int32_t h, int32_t w,
                                         int32_t channels, int32_t reverse, void *stream) {
    MVX_CHECK_ARG(in && out && n_frames >= 1 && planes >= 1 && h > 0 && w > 0 && h % 2 == 0 && w % 2 == 0);
    MVX_CHECK_ARG(channels > 0 && channels % 4 == 0);
    const size_t total = (size_t)n_frames * planes * h * w * (channels / 4);
    // reverse: `in` is the space-to-depth image and `out` the full-resolution tensor
    hipLaunchKernelGGL(space_to_depth, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, in, out, n_frames, planes, h, w,
                       channels, reverse);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" int mvx_d2s_bn_apply_frames(float *t, const float *mean_inv, float *out, int32_t n_frames, int32_t h, int32_t w,
                                       int32_t s, int32_t channels, int32_t ld_out, int32_t col_offset, int32_t reverse,
                                       void *stream) {
    MVX_CHECK_ARG(t && out && n_frames >= 1 && h > 0 && w > 0 && s >= 1 && channels > 0 && channels % 4 == 0);
    MVX_CHECK_ARG(ld_out % 4 == 0 && col_offset % 4 == 0 && col_offset >= 0 && col_offset + channels <= ld_out);
    const size_t total = (size_t)n_frames * h * w * s * s * (channels / 4);
    hipLaunchKernelGGL(d2s_rows, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, t, mean_inv, out, n_frames, h, w, s,
                       channels, ld_out, col_offset, reverse);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" int mvx_bn_apply_strided_frames(float *y, const float *mean_inv, float *out, int64_t rows, int32_t channels,
                                           int32_t ld_out, int32_t col_offset, int32_t n_frames, int32_t reverse,
                                           void *stream) {
    MVX_CHECK_ARG(y && out && rows >= 0 && channels > 0 && channels % 4 == 0 && n_frames >= 1 && rows % n_frames == 0);
    MVX_CHECK_ARG(ld_out % 4 == 0 && col_offset % 4 == 0 && col_offset >= 0 && col_offset + channels <= ld_out);
    if (rows == 0) return MVX_OK;
    hipLaunchKernelGGL(bn_apply_strided, dim3(ew_grid((size_t)rows * channels / 4)), dim3(256), 0, (hipStream_t)stream, y,
                       mean_inv, out, (size_t)rows, (size_t)(rows / n_frames), channels, ld_out, col_offset, reverse);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" int mvx_row_stats_frames(const float *y, double *stats, int64_t rows, int32_t channels, int32_t n_frames,
                                    void *stream) {
    MVX_CHECK_ARG(y && stats && rows >= 0 && channels > 0 && channels % 4 == 0 && channels <= 1024);
    MVX_CHECK_ARG(n_frames >= 1 && n_frames <= MVX_MAX_FRAMES && rows % n_frames == 0);
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(stats, 0, sizeof(double) * MVX_REP * 2 * channels * n_frames, st);
    if (e != hipSuccess) return (int)e;
    if (rows == 0) return MVX_OK;
    const size_t per = (size_t)(rows / n_frames);
    const int rpi = (256 / (channels / 4)) > 1 ? 256 / (channels / 4) : 1;
    size_t b = (per + (size_t)rpi - 1) / rpi;
    if (b > 512) b = 512;
    hipLaunchKernelGGL(row_stats_frames, dim3((unsigned)b, n_frames), dim3(256), 0, st, y, stats, per, channels);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}
