// Sparse voxel features <-> dense channels-last grid.
//
// Forward = the reference's VoxelNet.reindex (modules/voxelnet/VoxelNet.py:16-22):
//   res[0, :, iz, ix, iy] = x[v, :]   with idx rows (b, ix, iy, iz)
// The dense grid is kept channels-last, [D=iz][H=ix][W=iy][C], so each voxel is ONE contiguous
// C*4-byte row (512 B at C=128) instead of C separate cache lines in NCDHW.
// Backward = gather of the same rows from the grid gradient.
#include "common.h"

namespace {

__global__ void scatter_rows(const float *__restrict__ feat, const long long *__restrict__ coords,
                             float *__restrict__ grid, int V, int C, int D, int H, int W, int *status,
                             int *__restrict__ occ, int tile_h, int tile_w, unsigned *__restrict__ bits) {
    const int c4 = C >> 2;
    const size_t total = (size_t)V * c4;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int v = (int)(e / c4), part = (int)(e % c4);
        const long long ix = coords[(size_t)v * 4 + 1], iy = coords[(size_t)v * 4 + 2], iz = coords[(size_t)v * 4 + 3];
        if (ix < 0 || ix >= H || iy < 0 || iy >= W || iz < 0 || iz >= D) {
            if (status) atomicOr(status, 1);
            continue;
        }
        const size_t site = ((size_t)iz * H + ix) * W + iy;
        *(float4 *)(grid + site * C + part * 4) = *(const float4 *)(feat + (size_t)v * C + part * 4);
        if (occ && part == 0) {
            const int ty = (H + tile_h - 1) / tile_h, tx = (W + tile_w - 1) / tile_w;
            atomicAdd(&occ[((size_t)iz * ty + ix / tile_h) * tx + iy / tile_w], 1);
        }
        if (bits && part == 0) atomicOr(&bits[((size_t)iz * H + ix) * ((W + 31) / 32) + (iy >> 5)], 1u << (iy & 31));
    }
}

__global__ void gather_rows(const float *__restrict__ grid, const long long *__restrict__ coords,
                            float *__restrict__ feat, int V, int C, int D, int H, int W) {
    const int c4 = C >> 2;
    const size_t total = (size_t)V * c4;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int v = (int)(e / c4), part = (int)(e % c4);
        const long long ix = coords[(size_t)v * 4 + 1], iy = coords[(size_t)v * 4 + 2], iz = coords[(size_t)v * 4 + 3];
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ix >= 0 && ix < H && iy >= 0 && iy < W && iz >= 0 && iz < D) {
            const size_t site = ((size_t)iz * H + ix) * W + iy;
            o = *(const float4 *)(grid + site * C + part * 4);
        }
        *(float4 *)(feat + (size_t)v * C + part * 4) = o;
    }
}

}  // namespace

extern "C" int mvx_scatter_voxels(const float *feat, const int64_t *coords, float *grid, int32_t n_voxels,
                                  int32_t channels, int32_t d, int32_t h, int32_t w, int32_t zero_grid,
                                  int32_t *status, int32_t *occupancy, int32_t tile_h, int32_t tile_w,
                                  uint32_t *site_bits, void *stream) {
    MVX_CHECK_ARG(grid && channels > 0 && channels % 4 == 0 && d > 0 && h > 0 && w > 0 && n_voxels >= 0);
    hipStream_t st = (hipStream_t)stream;
    if (zero_grid) {
        hipError_t e = hipMemsetAsync(grid, 0, (size_t)d * h * w * channels * sizeof(float), st);
        if (e != hipSuccess) return (int)e;
    }
    if (occupancy) {
        MVX_CHECK_ARG(tile_h > 0 && tile_w > 0);
        hipError_t e = hipMemsetAsync(occupancy, 0, sizeof(int32_t) * (size_t)d * mvx_cdiv(h, tile_h) * mvx_cdiv(w, tile_w), st);
        if (e != hipSuccess) return (int)e;
    }
    if (site_bits) {
        hipError_t e = hipMemsetAsync(site_bits, 0, sizeof(uint32_t) * (size_t)d * h * mvx_cdiv(w, 32), st);
        if (e != hipSuccess) return (int)e;
    }
    if (n_voxels == 0) return MVX_OK;
    MVX_CHECK_ARG(feat && coords);
    const size_t total = (size_t)n_voxels * (channels / 4);
    hipLaunchKernelGGL(scatter_rows, dim3(mvx_cdiv(total, 256) > 4096 ? 4096 : mvx_cdiv(total, 256)), dim3(256), 0, st,
                       feat, (const long long *)coords, grid, n_voxels, channels, d, h, w, status, occupancy, tile_h, tile_w, site_bits);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" int mvx_gather_voxels(const float *grid, const int64_t *coords, float *feat, int32_t n_voxels,
                                 int32_t channels, int32_t d, int32_t h, int32_t w, void *stream) {
    MVX_CHECK_ARG(grid && channels > 0 && channels % 4 == 0 && d > 0 && h > 0 && w > 0 && n_voxels >= 0);
    if (n_voxels == 0) return MVX_OK;
    MVX_CHECK_ARG(feat && coords);
    const size_t total = (size_t)n_voxels * (channels / 4);
    hipLaunchKernelGGL(gather_rows, dim3(mvx_cdiv(total, 256) > 4096 ? 4096 : mvx_cdiv(total, 256)), dim3(256), 0,
                       (hipStream_t)stream, grid, (const long long *)coords, feat, n_voxels, channels, d, h, w);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

// ------------------------------------------------------------------------------------------
// CML output <-> bird's-eye-view map.  The reference reshapes the NCDHW result (1,64,2,H,W) to
// (1,128,H,W) (modules/voxelnet/VoxelNet.py:36): BEV channel = c*D + d.  With channels-last
// storage [D][H][W][C] this is a transposition, done here through a padded LDS tile so both the
// reads (C contiguous) and the writes (W contiguous) are coalesced.
// ------------------------------------------------------------------------------------------
namespace {

// cl [D][H][W][C]  ->  bev [C*D][H][W]   (dir = 0)   or back (dir = 1)
// copies > 1 (dir = 1 only): the ONE (C*D, H, W) map in src is written into `copies` frames of dst (a gradient shared by
// all frames of a step: one read, `copies` writes, instead of a transposition and a repeat of its result)
__global__ __launch_bounds__(256) void cl_bev_transpose(const float *__restrict__ src, float *__restrict__ dst,
                                                        int D, int H, int W, int C, int dir, int copies) {
    __shared__ float tile[32][33];
    const int w0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    // blockIdx.z runs over (frame, d, h): both tensors hold the frames back to back
    const int fr = blockIdx.z / (D * H), dh = blockIdx.z - fr * D * H, d = dh / H, h = dh % H;
    if (dir == 0) { src += (size_t)fr * D * H * W * C; dst += (size_t)fr * C * D * H * W; }
    else { src += (size_t)fr * C * D * H * W; dst += (size_t)fr * D * H * W * C; }
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    if (dir == 0) {
        for (int j = ty; j < 32; j += 8) {
            const int w = w0 + j, c = c0 + tx;
            tile[j][tx] = (w < W && c < C) ? src[(((size_t)d * H + h) * W + w) * C + c] : 0.f;
        }
        __syncthreads();
        for (int j = ty; j < 32; j += 8) {
            const int c = c0 + j, w = w0 + tx;
            if (w < W && c < C) dst[(((size_t)c * D + d) * H + h) * W + w] = tile[tx][j];
        }
    } else {
        for (int j = ty; j < 32; j += 8) {
            const int c = c0 + j, w = w0 + tx;
            tile[j][tx] = (w < W && c < C) ? src[(((size_t)c * D + d) * H + h) * W + w] : 0.f;
        }
        __syncthreads();
        for (int j = ty; j < 32; j += 8) {
            const int w = w0 + j, c = c0 + tx;
            if (w < W && c < C)
                for (int k = 0; k < copies; ++k) dst[(size_t)k * D * H * W * C + (((size_t)d * H + h) * W + w) * C + c] = tile[tx][j];
        }
    }
}

}  // namespace

extern "C" int mvx_cl_to_bev(const float *cl, float *bev, int32_t d, int32_t h, int32_t w, int32_t channels,
                             int32_t reverse, void *stream) {
    return mvx_cl_to_bev_frames(cl, bev, d, h, w, channels, reverse, 1, stream);
}

extern "C" int mvx_cl_to_bev_frames(const float *cl, float *bev, int32_t d, int32_t h, int32_t w, int32_t channels,
                                    int32_t reverse, int32_t n_frames, void *stream) {
    MVX_CHECK_ARG(cl && bev && d > 0 && h > 0 && w > 0 && channels > 0 && n_frames >= 1 && n_frames <= MVX_MAX_FRAMES);
    MVX_CHECK_ARG((long long)d * h * n_frames <= 65535);
    const dim3 grid(mvx_cdiv(w, 32), mvx_cdiv(channels, 32), d * h * n_frames);
    if (!reverse)
        hipLaunchKernelGGL(cl_bev_transpose, grid, dim3(256), 0, (hipStream_t)stream, cl, bev, d, h, w, channels, 0, 1);
    else
        hipLaunchKernelGGL(cl_bev_transpose, grid, dim3(256), 0, (hipStream_t)stream, (const float *)bev, (float *)cl, d,
                           h, w, channels, 1, 1);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

// One (C*D, H, W) map -> `copies` channels-last frames [copies][D][H][W][C]: a gradient of the middle output shared by all frames
// of a step (modules/pipeline.py: grad_mid of shape (1, 128, H, W)) in its channels-last form for every frame, in one pass
extern "C" int mvx_bev_to_cl_broadcast(const float *bev, float *cl, int32_t d, int32_t h, int32_t w, int32_t channels,
                                       int32_t copies, void *stream) {
    MVX_CHECK_ARG(cl && bev && d > 0 && h > 0 && w > 0 && channels > 0 && copies >= 1 && copies <= MVX_MAX_FRAMES);
    MVX_CHECK_ARG((long long)d * h <= 65535);
    const dim3 grid(mvx_cdiv(w, 32), mvx_cdiv(channels, 32), d * h);
    hipLaunchKernelGGL(cl_bev_transpose, grid, dim3(256), 0, (hipStream_t)stream, bev, cl, d, h, w, channels, 1, copies);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}
