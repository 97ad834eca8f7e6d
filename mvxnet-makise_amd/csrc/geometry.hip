// Point-cloud geometry on the GPU: range crop, camera-frustum crop and lidar -> image projection.
//
// Replaces modules/data/Preprocessing.py:12-55 (`crop`, `cropTensor`, `cropToSight`) and
// modules/utils/Calib.py:47-69 (`lidar2Img`).  The crops are order-preserving stream compactions
// (count / scan / write, one pass over the raw cloud each); both filters can run fused in one call
// because `cropToSight(crop(x))` keeps exactly the points that pass both masks, in order.
//
// Arithmetic follows the path being replaced: the numpy path compares f32 coordinates promoted to
// f64 against f64 bounds and projects in f64 (calib is f64 there, cropdata.py:46-56); the torch
// path rounds the bounds to f32 and projects in f32 (`math_f32`).
#include "common.h"

namespace {

struct CropParams {
    double lo[3], hi[3];     // range crop
    double m[16];            // R0_rect @ Tr_velo_to_cam, row-major 4x4
    double p2[16];           // P2, row-major 4x4
    double lim_w, lim_h;     // imsize (w, h) - 1e-3
    int use_range, use_sight, math_f32;
};

template <typename T>
__device__ __forceinline__ void project(const CropParams &c, float x, float y, float z, T *camz, T *u, T *v) {
    T cam[4], img[3];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        cam[i] = ((T)c.m[4 * i] * (T)x + (T)c.m[4 * i + 1] * (T)y) + ((T)c.m[4 * i + 2] * (T)z + (T)c.m[4 * i + 3]);
#pragma unroll
    for (int i = 0; i < 3; ++i)
        img[i] = ((T)c.p2[4 * i] * cam[0] + (T)c.p2[4 * i + 1] * cam[1]) +
                 ((T)c.p2[4 * i + 2] * cam[2] + (T)c.p2[4 * i + 3] * cam[3]);
    *camz = cam[2];
    *u = img[0] / img[2];
    *v = img[1] / img[2];
}

__device__ __forceinline__ bool keep_point(const CropParams &c, const float *p) {
    bool ok = true;
    if (c.use_range) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double v = (double)p[a];
            ok = ok && (c.lo[a] <= v) && (v < c.hi[a]);
        }
    }
    if (ok && c.use_sight) {
        if (c.math_f32) {
            float z, u, v;
            project<float>(c, p[0], p[1], p[2], &z, &u, &v);
            ok = (z > 0.f) && (u >= 0.f) && (v >= 0.f) && (u < (float)c.lim_w) && (v < (float)c.lim_h);
        } else {
            double z, u, v;
            project<double>(c, p[0], p[1], p[2], &z, &u, &v);
            ok = (z > 0.0) && (u >= 0.0) && (v >= 0.0) && (u < c.lim_w) && (v < c.lim_h);
        }
    }
    return ok;
}

__global__ __launch_bounds__(256) void crop_count(const float *__restrict__ pcd, const int *__restrict__ n_in, int cap,
                                                  int ncol, CropParams c, int *__restrict__ bcount, int nblocks) {
    __shared__ int s[4];
    const int f = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int n = min(n_in ? n_in[f] : cap, cap);
    const int flag = i < n && keep_point(c, pcd + ((size_t)f * cap + i) * ncol);
    const unsigned long long b = __ballot(flag);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = __popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) bcount[(size_t)f * nblocks + blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

__global__ __launch_bounds__(1024) void crop_scan(int *__restrict__ bcount, int nblocks, int *__restrict__ n_out) {
    __shared__ int smem[17];
    const int f = blockIdx.x;
    int *bc = bcount + (size_t)f * nblocks;
    int base = 0;
    for (int t0 = 0; t0 < nblocks; t0 += 1024) {
        const int i = t0 + threadIdx.x;
        const int v = i < nblocks ? bc[i] : 0;
        int tot;
        const int ex = block_excl_scan_i32(v, smem, &tot);
        if (i < nblocks) bc[i] = base + ex;
        base += tot;
    }
    if (threadIdx.x == 0) n_out[f] = base;
}

// has_proj: the kept row is written as [its ncol columns, row, col] with the f32 projection of train.py:31-34
// (lidar2Img on the torch path, swapped to (row, col)); out rows have ncol_out columns and frame stride cap_out.
__global__ __launch_bounds__(256) void crop_write(const float *__restrict__ pcd, const int *__restrict__ n_in, int cap,
                                                  int ncol, CropParams c, const int *__restrict__ boff, int nblocks,
                                                  float *__restrict__ out, int *__restrict__ src_index, int ncol_out,
                                                  int cap_out, int has_proj, CropParams proj) {
    __shared__ int s[4];
    const int f = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n = min(n_in ? n_in[f] : cap, cap);
    const float *p = pcd + ((size_t)f * cap + i) * ncol;
    const int flag = i < n && keep_point(c, p);
    const unsigned long long b = __ballot(flag);
    if (lane == 0) s[wv] = __popcll(b);
    __syncthreads();
    int off = boff[(size_t)f * nblocks + blockIdx.x];
    for (int k = 0; k < wv; ++k) off += s[k];
    off += __popcll(b & ((1ull << lane) - 1ull));
    if (flag && off < cap_out) {
        float *o = out + ((size_t)f * cap_out + off) * ncol_out;
        for (int a = 0; a < ncol; ++a) o[a] = p[a];
        if (has_proj) {
            float z, u, v;
            project<float>(proj, p[0], p[1], p[2], &z, &u, &v);
            o[ncol] = v;                  // (row, col) = (v, u): train.py:33
            o[ncol + 1] = u;
        }
        if (src_index) src_index[(size_t)f * cap_out + off] = i;
    }
}

__global__ void lidar2img(const float *__restrict__ pcd, int ncol, long long n, CropParams c, float *__restrict__ out,
                          int ldo, int col_off, int swap_rc, float *__restrict__ cam_z) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *p = pcd + i * ncol;
    float u, v, z;
    if (c.math_f32) {
        project<float>(c, p[0], p[1], p[2], &z, &u, &v);
    } else {
        double zd, ud, vd;
        project<double>(c, p[0], p[1], p[2], &zd, &ud, &vd);
        z = (float)zd; u = (float)ud; v = (float)vd;
    }
    float *o = out + i * ldo + col_off;
    o[0] = swap_rc ? v : u;
    o[1] = swap_rc ? u : v;
    if (cam_z) cam_z[i] = z;
}

inline void fill_params(CropParams &c, const double *range6, int bounds_f32, const double *m_host, const double *p2_host,
                        double w, double h, int math_f32) {
    c.use_range = range6 != nullptr;
    c.use_sight = m_host != nullptr;
    c.math_f32 = math_f32;
    for (int a = 0; a < 3; ++a) {
        c.lo[a] = range6 ? (bounds_f32 ? (double)(float)range6[a] : range6[a]) : 0.0;
        c.hi[a] = range6 ? (bounds_f32 ? (double)(float)range6[a + 3] : range6[a + 3]) : 0.0;
    }
    for (int k = 0; k < 16; ++k) {
        c.m[k] = m_host ? m_host[k] : 0.0;
        c.p2[k] = p2_host ? p2_host[k] : 0.0;
    }
    // Preprocessing.py:37,41: imsize - 1e-3, in f64 (numpy) or f32 (torch)
    c.lim_w = math_f32 ? (double)((float)w - 1e-3f) : w - 1e-3;
    c.lim_h = math_f32 ? (double)((float)h - 1e-3f) : h - 1e-3;
}

}  // namespace

extern "C" size_t mvx_crop_workspace_bytes(int32_t n_frames, int32_t cap_points) {
    if (n_frames <= 0 || cap_points <= 0) return 0;
    return (size_t)n_frames * mvx_cdiv(cap_points, 256) * sizeof(int32_t);
}

extern "C" int mvx_crop_points(const float *pcd, const int32_t *n_in, int32_t n_frames, int32_t cap_points,
                               int32_t ncol, const double *range6_host, int32_t bounds_f32,
                               const double *cam_from_velo_host, const double *p2_host, double imsize_w,
                               double imsize_h, int32_t math_f32, float *out, int32_t *n_out, int32_t *src_index,
                               void *workspace, size_t workspace_bytes, void *stream) {
    MVX_CHECK_ARG(pcd && out && n_out && workspace && n_frames > 0 && cap_points > 0 && ncol >= 3);
    MVX_CHECK_ARG(range6_host || cam_from_velo_host);
    MVX_CHECK_ARG(!cam_from_velo_host || p2_host);
    MVX_CHECK_ARG(workspace_bytes >= mvx_crop_workspace_bytes(n_frames, cap_points));
    CropParams c;
    fill_params(c, range6_host, bounds_f32, cam_from_velo_host, p2_host, imsize_w, imsize_h, math_f32);
    hipStream_t st = (hipStream_t)stream;
    const int nb = (int)mvx_cdiv(cap_points, 256);
    int *bc = (int *)workspace;
    hipLaunchKernelGGL(crop_count, dim3(nb, n_frames), dim3(256), 0, st, pcd, n_in, cap_points, ncol, c, bc, nb);
    MVX_LAUNCH_CHECK();
    hipLaunchKernelGGL(crop_scan, dim3(n_frames), dim3(1024), 0, st, bc, nb, n_out);
    MVX_LAUNCH_CHECK();
    hipLaunchKernelGGL(crop_write, dim3(nb, n_frames), dim3(256), 0, st, pcd, n_in, cap_points, ncol, c, (const int *)bc,
                       nb, out, src_index, ncol, cap_points, 0, c);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" size_t mvx_crop_project_workspace_bytes(int32_t n_frames, int32_t cap_points) {
    return mvx_crop_workspace_bytes(n_frames, cap_points);
}

extern "C" int mvx_crop_project_points(const float *pcd, const int32_t *n_in, int32_t n_frames, int32_t cap_points,
                                       int32_t ncol, const double *range6_host, const double *cam_from_velo_host,
                                       const double *p2_host, double imsize_w, double imsize_h,
                                       const double *proj_cam_from_velo_host, const double *proj_p2_host, float *out,
                                       int32_t cap_out, int32_t *n_out, void *workspace, size_t workspace_bytes,
                                       void *stream) {
    MVX_CHECK_ARG(pcd && out && n_out && workspace && n_frames > 0 && cap_points > 0 && cap_out > 0 && ncol >= 3);
    MVX_CHECK_ARG(range6_host && cam_from_velo_host && p2_host && proj_cam_from_velo_host && proj_p2_host);
    MVX_CHECK_ARG(workspace_bytes >= mvx_crop_project_workspace_bytes(n_frames, cap_points));
    CropParams c, pj;
    fill_params(c, range6_host, 0, cam_from_velo_host, p2_host, imsize_w, imsize_h, 0);        // numpy-path masks (f64)
    fill_params(pj, nullptr, 0, proj_cam_from_velo_host, proj_p2_host, 0.0, 0.0, 1);            // torch-path projection (f32)
    hipStream_t st = (hipStream_t)stream;
    const int nb = (int)mvx_cdiv(cap_points, 256);
    int *bc = (int *)workspace;
    hipLaunchKernelGGL(crop_count, dim3(nb, n_frames), dim3(256), 0, st, pcd, n_in, cap_points, ncol, c, bc, nb);
    MVX_LAUNCH_CHECK();
    hipLaunchKernelGGL(crop_scan, dim3(n_frames), dim3(1024), 0, st, bc, nb, n_out);
    MVX_LAUNCH_CHECK();
    hipLaunchKernelGGL(crop_write, dim3(nb, n_frames), dim3(256), 0, st, pcd, n_in, cap_points, ncol, c, (const int *)bc,
                       nb, out, (int *)nullptr, ncol + 2, cap_out, 1, pj);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" int mvx_lidar2img(const float *pcd, int32_t ncol, int64_t n_points, const double *cam_from_velo_host,
                             const double *p2_host, int32_t math_f32, float *out, int32_t ld_out, int32_t col_offset,
                             int32_t swap_to_row_col, float *cam_z, void *stream) {
    MVX_CHECK_ARG(pcd && out && cam_from_velo_host && p2_host && ncol >= 3 && n_points >= 0);
    MVX_CHECK_ARG(ld_out >= col_offset + 2 && col_offset >= 0);
    if (n_points == 0) return MVX_OK;
    CropParams c;
    fill_params(c, nullptr, 0, cam_from_velo_host, p2_host, 0.0, 0.0, math_f32);
    hipLaunchKernelGGL(lidar2img, dim3(mvx_cdiv(n_points, 256)), dim3(256), 0, (hipStream_t)stream, pcd, ncol,
                       (long long)n_points, c, out, ld_out, col_offset, swap_to_row_col, cam_z);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}
