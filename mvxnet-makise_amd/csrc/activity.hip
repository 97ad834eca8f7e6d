// Activity maps and background constants of the CML convolution stack.
//
// The grid VoxelNet.reindex fills (modules/voxelnet/VoxelNet.py:16-22) is zero outside the V voxel
// sites, and every block of CML (modules/voxelnet/Pipe.py:31-43) is Conv3d -> ReLU -> BatchNorm
// without affine (modules/layers/Blocks.py:20-29).  A site whose whole receptive field holds no voxel
// therefore carries, after each layer, ONE value per channel and depth plane -- ReLU(bias) after the
// first convolution, ReLU(bias + sum_taps W * c_prev) after the next ones -- the same at every such
// "background" site, except where the 3x3 window leaves the image (zero padding is not the background).
// These kernels find the background sites (exact dilation of the voxel occupancy, layer by layer) and
// evaluate the constants, so that the convolution kernels can
//   - forward: fill background tiles with the constant instead of convolving them,
//   - wgrad  : sum (x - c) (x) dz over the tiles that hold a non-background site only and add the
//              remaining c (x) sum(dz) term in closed form,
// both exact rewrites of the dense arithmetic (differences at fp32 rounding level).
#include "common.h"

namespace {

constexpr int ATH = 8, ATW = 16;        // tile of the convolution kernels (conv3d.hip TH x TW)

// dst[d][y][x] = 1 iff any source site in the 3x3x3 receptive field is active, or (mark_border and the
// in-plane window leaves the image).  Source: int32 index grid (voxel id, -1 = empty) or uint8 mask.
__global__ void activity_sites(const void *__restrict__ src, int src_is_index, int Din, int Dout, int H, int W, int sd,
                               int pd, int mark_border, unsigned char *__restrict__ dst) {
    const size_t n = (size_t)Dout * H * W;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(e % W), y = (int)((e / W) % H), d = (int)(e / ((size_t)W * H));
        int on = mark_border && (y == 0 || y == H - 1 || x == 0 || x == W - 1);
        for (int kd = 0; kd < 3 && !on; ++kd) {
            const int ds = d * sd - pd + kd;
            if (ds < 0 || ds >= Din) continue;
            for (int a = -1; a <= 1 && !on; ++a) {
                const int yy = y + a;
                if (yy < 0 || yy >= H) continue;
                for (int b = -1; b <= 1; ++b) {
                    const int xx = x + b;
                    if (xx < 0 || xx >= W) continue;
                    const size_t s = ((size_t)ds * H + yy) * W + xx;
                    on |= src_is_index ? (((const int *)src)[s] >= 0) : (((const unsigned char *)src)[s] != 0);
                }
            }
        }
        dst[e] = (unsigned char)on;
    }
}

// flags[d][tile] = 1 iff the (TH+2) x (TW+2) halo of the tile holds an active site of plane d
__global__ __launch_bounds__(256) void activity_halo_flags(const unsigned char *__restrict__ mask, int D, int H, int W,
                                                           int *__restrict__ flags) {
    const int tiles_x = (W + ATW - 1) / ATW;
    const int tx0 = (blockIdx.x % tiles_x) * ATW - 1, ty0 = (blockIdx.x / tiles_x) * ATH - 1;
    const int d = blockIdx.y;
    int on = 0;
    if (threadIdx.x < (ATH + 2) * (ATW + 2)) {
        const int y = ty0 + threadIdx.x / (ATW + 2), x = tx0 + threadIdx.x % (ATW + 2);
        if (y >= 0 && y < H && x >= 0 && x < W) on = mask[((size_t)d * H + y) * W + x];
    }
    on = __syncthreads_or(on);
    if (threadIdx.x == 0) flags[(size_t)d * gridDim.x + blockIdx.x] = on ? 1 : 0;
}

// bg_pre[d][n] = sum over the valid depth taps of plane d and all in-plane taps / channels of
//                W[n][c][kd][a][b] * c_in[src(d,kd)][c]            (f64 accumulation, rounded once)
__global__ void conv_background(const float *__restrict__ w, const float *__restrict__ c_in, int Din, int Dout, int Cin,
                                int Cout, int sd, int pd, float *__restrict__ bg_pre) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, d = blockIdx.y;
    if (n >= Cout) return;
    double s = 0.0;
    for (int kd = 0; kd < 3; ++kd) {
        const int ds = d * sd - pd + kd;
        if (ds < 0 || ds >= Din) continue;
        for (int c = 0; c < Cin; ++c) {
            const float *wp = w + (((size_t)n * Cin + c) * 3 + kd) * 9;
            double t = 0.0;
            for (int k = 0; k < 9; ++k) t += (double)wp[k];
            s += t * (double)c_in[(size_t)ds * Cin + c];
        }
    }
    bg_pre[(size_t)d * Cout + n] = (float)s;
}

// y_bg = [ReLU](bg_pre + bias) and c_out = (y_bg - mean) * inv, with exactly the fp32 operations of the
// convolution epilogue and of bn_apply, so that c_out equals the normalised tensor at background sites bit for bit.
__global__ void bn_background(const float *__restrict__ bg_pre, const float *__restrict__ bias, const float *__restrict__ mi,
                              int D, int C, int relu, float *__restrict__ y_bg, float *__restrict__ c_out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= D * C) return;
    const int c = e % C;
    float v = (bg_pre ? bg_pre[e] : 0.f) + (bias ? bias[c] : 0.f);
    if (relu) v = fmaxf(v, 0.f);
    if (y_bg) y_bg[e] = v;
    c_out[e] = (v - mi[c]) * mi[C + c];
}

}  // namespace

extern "C" int mvx_activity_dilate(const void *src, int32_t src_is_index, int32_t din, int32_t dout, int32_t h, int32_t w,
                                   int32_t stride_d, int32_t pad_d, int32_t mark_border, uint8_t *dst_mask,
                                   int32_t *dst_halo_flags, void *stream) {
    MVX_CHECK_ARG(src && dst_mask && din > 0 && dout > 0 && h > 0 && w > 0);
    MVX_CHECK_ARG(stride_d >= 1 && stride_d <= 2 && pad_d >= 0 && pad_d <= 1);
    MVX_CHECK_ARG(dout == (din + 2 * pad_d - 3) / stride_d + 1);
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)dout * h * w;
    hipLaunchKernelGGL(activity_sites, dim3(mvx_cdiv(n, 256) > 4096 ? 4096 : mvx_cdiv(n, 256)), dim3(256), 0, st, src,
                       src_is_index, din, dout, h, w, stride_d, pad_d, mark_border, dst_mask);
    MVX_LAUNCH_CHECK();
    if (dst_halo_flags) {
        hipLaunchKernelGGL(activity_halo_flags, dim3(mvx_cdiv(w, ATW) * mvx_cdiv(h, ATH), dout), dim3(256), 0, st,
                           (const unsigned char *)dst_mask, dout, h, w, dst_halo_flags);
        MVX_LAUNCH_CHECK();
    }
    return MVX_OK;
}

extern "C" int mvx_conv3d_background(const float *w, const float *c_in, int32_t din, int32_t dout, int32_t cin,
                                     int32_t cout, int32_t stride_d, int32_t pad_d, float *bg_pre, void *stream) {
    MVX_CHECK_ARG(w && c_in && bg_pre && din > 0 && dout > 0 && cin > 0 && cout > 0);
    hipLaunchKernelGGL(conv_background, dim3(mvx_cdiv(cout, 64), dout), dim3(64), 0, (hipStream_t)stream, w, c_in, din, dout,
                       cin, cout, stride_d, pad_d, bg_pre);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" int mvx_bn_background(const float *bg_pre, const float *bias, const float *mean_inv, int32_t planes,
                                 int32_t channels, int32_t flags, float *y_bg, float *c_out, void *stream) {
    MVX_CHECK_ARG(mean_inv && c_out && planes > 0 && channels > 0);
    hipLaunchKernelGGL(bn_background, dim3(mvx_cdiv((long long)planes * channels, 256)), dim3(256), 0, (hipStream_t)stream,
                       bg_pre, bias, mean_inv, planes, channels, flags & MVX_FLAG_RELU, y_bg, c_out);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}
