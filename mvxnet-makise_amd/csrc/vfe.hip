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
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void vfe_bn_max_concat(const float *__restrict__ y, const float *__restrict__ mi,
                                                         float *__restrict__ out, int *__restrict__ argmax,
                                                         int V, int T, int C) {
    const long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (e >= (long long)V * C) return;
    const int v = (int)(e / C), c = (int)(e % C);
    const float m = mi[c], iv = mi[C + c];
    const float *src = y + (size_t)v * T * C + c;
    float best = -INFINITY;
    int bi = 0;
    for (int t = 0; t < T; ++t) {
        const float val = (src[(size_t)t * C] - m) * iv;
        if (val > best) { best = val; bi = t; }
    }
    float *dst = out + (size_t)v * T * 2 * C + c;
    for (int t = 0; t < T; ++t) {
        dst[(size_t)t * 2 * C] = (src[(size_t)t * C] - m) * iv;
        dst[(size_t)t * 2 * C + C] = best;
    }
    argmax[e] = bi;
}

__global__ __launch_bounds__(256) void vfe_max_concat_bwd(const float *__restrict__ g, const int *__restrict__ argmax,
                                                          float *__restrict__ dyh, int V, int T, int C) {
    const long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (e >= (long long)V * C) return;
    const int v = (int)(e / C), c = (int)(e % C);
    const float *src = g + (size_t)v * T * 2 * C + c;
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += src[(size_t)t * 2 * C + C];
    const int am = argmax[e];
    float *dst = dyh + (size_t)v * T * C + c;
    for (int t = 0; t < T; ++t) dst[(size_t)t * C] = src[(size_t)t * 2 * C] + (t == am ? s : 0.f);
}

__global__ __launch_bounds__(256) void bn_segmax(const float *__restrict__ y, const float *__restrict__ mi,
                                                 float *__restrict__ out, int *__restrict__ argmax, int V, int T, int C) {
    const long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (e >= (long long)V * C) return;
    const int v = (int)(e / C), c = (int)(e % C);
    const float m = mi[c], iv = mi[C + c];
    const float *src = y + (size_t)v * T * C + c;
    float best = -INFINITY;
    int bi = 0;
    for (int t = 0; t < T; ++t) {
        const float val = (src[(size_t)t * C] - m) * iv;
        if (val > best) { best = val; bi = t; }
    }
    out[e] = best;
    argmax[e] = bi;
}

__global__ __launch_bounds__(256) void segmax_bwd(const float *__restrict__ dfeat, const int *__restrict__ argmax,
                                                  float *__restrict__ dyh, int V, int T, int C) {
    const long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (e >= (long long)V * C) return;
    const int v = (int)(e / C), c = (int)(e % C);
    const float gval = dfeat[e];
    const int am = argmax[e];
    float *dst = dyh + (size_t)v * T * C + c;
    for (int t = 0; t < T; ++t) dst[(size_t)t * C] = (t == am) ? gval : 0.f;
}

}  // namespace

#define VFE_ARGS_OK (n_voxels >= 0 && t > 0 && channels > 0)
#define VFE_GRID dim3(mvx_cdiv((long long)n_voxels * channels, 256)), dim3(256), 0, (hipStream_t)stream

extern "C" int mvx_vfe_bn_max_concat(const float *y, const float *mean_inv, float *out, int32_t *argmax,
                                     int32_t n_voxels, int32_t t, int32_t channels, void *stream) {
    MVX_CHECK_ARG(y && mean_inv && out && argmax && VFE_ARGS_OK);
    if (n_voxels == 0) return MVX_OK;
    hipLaunchKernelGGL(vfe_bn_max_concat, VFE_GRID, y, mean_inv, out, argmax, n_voxels, t, channels);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" int mvx_vfe_max_concat_backward(const float *grad_out, const int32_t *argmax, float *dyhat,
                                           int32_t n_voxels, int32_t t, int32_t channels, void *stream) {
    MVX_CHECK_ARG(grad_out && argmax && dyhat && VFE_ARGS_OK);
    if (n_voxels == 0) return MVX_OK;
    hipLaunchKernelGGL(vfe_max_concat_bwd, VFE_GRID, grad_out, argmax, dyhat, n_voxels, t, channels);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" int mvx_bn_segment_max(const float *y, const float *mean_inv, float *out, int32_t *argmax,
                                  int32_t n_voxels, int32_t t, int32_t channels, void *stream) {
    MVX_CHECK_ARG(y && mean_inv && out && argmax && VFE_ARGS_OK);
    if (n_voxels == 0) return MVX_OK;
    hipLaunchKernelGGL(bn_segmax, VFE_GRID, y, mean_inv, out, argmax, n_voxels, t, channels);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" int mvx_segment_max_backward(const float *dfeat, const int32_t *argmax, float *dyhat, int32_t n_voxels,
                                        int32_t t, int32_t channels, void *stream) {
    MVX_CHECK_ARG(dfeat && argmax && dyhat && VFE_ARGS_OK);
    if (n_voxels == 0) return MVX_OK;
    hipLaunchKernelGGL(segmax_bwd, VFE_GRID, dfeat, argmax, dyhat, n_voxels, t, channels);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}
