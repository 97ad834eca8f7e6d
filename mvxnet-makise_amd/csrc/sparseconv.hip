// Input-sparse 3x3x3 convolution as voxel GEMMs + index-grid gathers.
//
// The first CML layer (modules/voxelnet/Pipe.py:36, Conv3d 128->64, stride (2,1,1), pad 1) reads the
// grid that VoxelNet.reindex (VoxelNet.py:16-22) just filled: zero except at the V voxel sites.  Its
// three passes therefore factor through a [V x 27*Cout] matrix:
//
//   forward : P = X W_all^T            (row GEMM, linear.hip; X = voxel rows [V][Cin],
//                                       W_all[(tap)*Cout + co][ci] = W[co][ci][kd][a][b])
//             out[site][co] = ReLU(b[co] + sum over the <=27 taps whose source site holds voxel v of
//                             P[v][tap*Cout + co])            -- `sparse_conv_output` below
//   dgrad   : G[v][tap*Cout + co] = dz[site that used v through tap][co]  -- `sparse_conv_gather_dz`
//             dX = G W_all             (row GEMM)
//   wgrad   : dW_all = G^T X           (row GEMM wgrad)
//
// The dense 721 MB input grid is never built: a 5.6 MB int32 index grid (voxel id or -1 per site)
// stands for it.  The output IS dense (every site gets ReLU(bias) at least), as the reference's.
// Sums run in a fixed order (k inside the GEMM, then taps in ascending order): deterministic, and
// equal to the dense convolution up to fp32 summation order.
#include "common.h"

namespace {

struct SGeom { int Din, Dout, H, W, Cout, sd, pd; };   // Din / Dout: planes PER FRAME (frames are stacked along depth)
constexpr int OTH = 8, OTW = 16;      // coarse occupancy tiles (sites)

__global__ void index_grid_fill(const long long *__restrict__ coords, int V, int D, int H, int W, int *__restrict__ grid,
                                int *__restrict__ occ, int *__restrict__ status, FrameMap fm) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    const long long ix = coords[(size_t)v * 4 + 1], iy = coords[(size_t)v * 4 + 2], iz = coords[(size_t)v * 4 + 3];
    if (ix < 0 || ix >= H || iy < 0 || iy >= W || iz < 0 || iz >= D) {
        if (status) atomicOr(status, 1);
        return;
    }
    const long long pz = (long long)fm_frame_of(fm, v) * D + iz;      // the voxel's frame owns planes [f*D, (f+1)*D)
    grid[((size_t)pz * H + ix) * W + iy] = v;
    const int ty = (H + OTH - 1) / OTH, tx = (W + OTW - 1) / OTW;
    atomicAdd(&occ[((size_t)pz * ty + ix / OTH) * tx + iy / OTW], 1);
}

// One workgroup = one 8 x 16-site occupancy tile of one output plane.  The coarse occupancy of the 3x3 tile neighbourhood is
// tested ONCE per workgroup: empty neighbourhoods (about 3/4 of the tiles on lidar frames) are a pure ReLU(bias) fill (or nothing:
// MVX_FLAG_NO_BG_FILL).  An occupied neighbourhood holds FEW voxels (7 (voxel, depth tap) pairs per tile on average), so the tile
// is built voxel by voxel: the voxel ids of the halo are compacted into a list in (depth tap, row, column) order, and for every
// entry nine groups of Cout/4 threads add the voxel's nine P rows of that depth tap into the nine output sites that read it, in an
// LDS accumulator tile.  A site therefore receives its terms in ascending (kd, a, b) order -- the order of the site-by-site
// form this replaces (27 conditional fetches per site, one memory latency per site row and depth tap: 0.29 ms per step), so the
// output and the BatchNorm sums are bit-identical to it.  The P rows of the next entries are in flight while one is added.
constexpr int SCO_MAXC = 64;            // Cout <= 64 (mvx_sparse_conv_output_frames checks Cout / 4 * OTW <= 256)
constexpr int SCO_AHEAD = 4;            // entries whose P rows are fetched before the first of them is added

__global__ __launch_bounds__(256) void sparse_conv_output(const float *__restrict__ P, const int *__restrict__ idx,
                                                          const int *__restrict__ occ,
                                                          const float *__restrict__ bias, float *__restrict__ out,
                                                          double *__restrict__ stats, SGeom g, int relu,
                                                          int *__restrict__ active_sites, int skip_fill,
                                                          const int *__restrict__ tile_flags) {
    __shared__ float red[2][256][4];
    __shared__ int s_idx[3 * (OTH + 2) * (OTW + 2)];   // voxel ids of the tile's halo, [depth tap][row][column]
    __shared__ int s_list[3 * (OTH + 2) * (OTW + 2)];  // halo positions that hold a voxel, ascending
    __shared__ int s_n;
    __shared__ __attribute__((aligned(16))) float s_acc[OTH * OTW * SCO_MAXC];
    constexpr int HP = (OTH + 2) * (OTW + 2), HN = 3 * HP;
    const int c4n = g.Cout >> 2;                       // threads per site (16 at Cout = 64)
    const int ct = threadIdx.x % c4n, st = threadIdx.x / c4n;       // st: site column inside the tile (0..15)
    const int tyn = (g.H + OTH - 1) / OTH, txn = (g.W + OTW - 1) / OTW;
    const int tcx = blockIdx.x % txn, tcy = blockIdx.x / txn, d = blockIdx.y;       // d: global output plane
    const int frame = d / g.Dout;
    if (stats) stats += (size_t)frame * MVX_REP * 2 * g.Cout;
    active_sites += frame;
    const int x = tcx * OTW + st;
    int any = 0;
    if (tile_flags) {
        // the caller knows the tiles of THIS output that hold a site with a voxel under its 27 taps (activity.hip: the tile flags of
        // the layer's output): one load, and fewer tiles than the coarse test below finds (it takes every neighbour of an
        // occupied tile).  A tile it leaves out holds ReLU(bias) at every site, which is what the closed form of
        // sparse_stats_fix counts for it.
        any = tile_flags[(size_t)d * (tyn * txn) + blockIdx.x];
    } else {
        for (int kd = 0; kd < 3; ++kd) {
            const int ds = mvx_src_plane(d, g.Din, g.Dout, g.sd, g.pd, kd);
            if (ds < 0) continue;
            for (int ty = max(tcy - 1, 0); ty <= min(tcy + 1, tyn - 1); ++ty)
                for (int tx = max(tcx - 1, 0); tx <= min(tcx + 1, txn - 1); ++tx) any |= occ[((size_t)ds * tyn + ty) * txn + tx];
        }
    }
    if (!any && skip_fill) return;                     // MVX_FLAG_NO_BG_FILL: the ReLU(bias) fill of a voxel-free tile is implied
    if (any) {                                         // block-uniform
        for (int e = threadIdx.x; e < HN; e += 256) {
            const int kd = e / HP, rem = e % HP;
            const int hy = rem / (OTW + 2), hx = rem % (OTW + 2);
            const int ds = mvx_src_plane(d, g.Din, g.Dout, g.sd, g.pd, kd), gy = tcy * OTH - 1 + hy, gx = tcx * OTW - 1 + hx;
            int v = -1;
            if (ds >= 0 && gy >= 0 && gy < g.H && gx >= 0 && gx < g.W) v = idx[((size_t)ds * g.H + gy) * g.W + gx];
            s_idx[e] = v;
        }
        for (int e = threadIdx.x; e < OTH * OTW * c4n; e += 256) *(float4 *)(s_acc + e * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
        if (threadIdx.x < 64) {                        // wave 0 compacts the halo in ascending position order
            int n = 0;
            for (int e0 = 0; e0 < HN; e0 += 64) {
                const int e = e0 + (int)threadIdx.x;
                const bool on = e < HN && s_idx[e] >= 0;
                const unsigned long long bal = __ballot(on);
                if (on) s_list[n + __popcll(bal & ((1ull << threadIdx.x) - 1ull))] = e;
                n += __popcll(bal);
            }
            if (threadIdx.x == 0) s_n = n;
        }
        __syncthreads();
        const int n = s_n;
        // thread -> (tap j of the 3 x 3 window, channel quad): the site it adds into depends on the entry
        const int j = threadIdx.x / c4n, a = j / 3, b = j % 3;
        const bool adder = j < 9;
        for (int i0 = 0; i0 < n; i0 += SCO_AHEAD) {
            float4 p[SCO_AHEAD];
            int site[SCO_AHEAD];
#pragma unroll
            for (int k = 0; k < SCO_AHEAD; ++k) {
                site[k] = -1;
                p[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (adder && i0 + k < n) {
                    const int e = s_list[i0 + k];
                    const int kd = e / HP, rem = e % HP;
                    const int r = rem / (OTW + 2) - a, c = rem % (OTW + 2) - b;       // the output site that reads it through (a, b)
                    if (r >= 0 && r < OTH && c >= 0 && c < OTW) {
                        site[k] = r * OTW + c;
                        p[k] = *(const float4 *)(P + ((size_t)s_idx[e] * 27 + (kd * 9 + j)) * g.Cout + ct * 4);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < SCO_AHEAD; ++k) {
                if (site[k] >= 0) {                    // the nine taps of one entry go to nine different sites
                    float4 *q = (float4 *)(s_acc + ((size_t)site[k] * c4n + ct) * 4);
                    float4 t = *q;
                    t.x += p[k].x; t.y += p[k].y; t.z += p[k].z; t.w += p[k].w;
                    *q = t;
                }
                __syncthreads();                       // the next entry may add into the same sites
            }
        }
    }
    float4 bs = bias ? *(const float4 *)(bias + ct * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool col_live = st < OTW && x < g.W;
    int live_sites = 0;
    for (int r = 0; r < OTH; ++r) {
        const int y = tcy * OTH + r;
        if (y >= g.H || !col_live) continue;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (any) acc = *(const float4 *)(s_acc + ((size_t)(r * OTW + st) * c4n + ct) * 4);
        acc.x += bs.x; acc.y += bs.y; acc.z += bs.z; acc.w += bs.w;
        if (relu) { acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f); }
        *(float4 *)(out + (((size_t)d * g.H + y) * g.W + x) * g.Cout + ct * 4) = acc;
        s1.x += acc.x; s1.y += acc.y; s1.z += acc.z; s1.w += acc.w;
        s2.x += acc.x * acc.x; s2.y += acc.y * acc.y; s2.z += acc.z * acc.z; s2.w += acc.w * acc.w;
        ++live_sites;
    }
    // Tiles without any source voxel wrote ReLU(bias) everywhere: their share of the BatchNorm sums
    // is added in closed form by sparse_stats_fix, so only the few active tiles touch the atomics.
    if (stats && any) {
        red[0][threadIdx.x][0] = s1.x; red[0][threadIdx.x][1] = s1.y; red[0][threadIdx.x][2] = s1.z; red[0][threadIdx.x][3] = s1.w;
        red[1][threadIdx.x][0] = s2.x; red[1][threadIdx.x][1] = s2.y; red[1][threadIdx.x][2] = s2.z; red[1][threadIdx.x][3] = s2.w;
        __syncthreads();
        const int spb = 256 / c4n;
        if (ct == 0) atomicAdd(active_sites, live_sites);
        if (st == 0) {
            const unsigned rep = (blockIdx.x + blockIdx.y * gridDim.x) % MVX_REP;
            for (int k = 0; k < 2; ++k)
                for (int j = 0; j < 4; ++j) {
                    double t = 0.0;
                    for (int q = 0; q < spb; ++q) t += (double)red[k][q * c4n + ct][j];
                    atomicAdd(stats + ((size_t)rep * 2 + k) * g.Cout + ct * 4 + j, t);
                }
        }
    }
}

__global__ void sparse_stats_fix(double *__restrict__ stats, const float *__restrict__ bias, const int *__restrict__ active_sites,
                                 long long total_sites, int C, int relu) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    stats += (size_t)blockIdx.y * MVX_REP * 2 * C;      // frame
    active_sites += blockIdx.y;
    double v = bias ? (double)bias[c] : 0.0;
    if (relu && v < 0.0) v = 0.0;
    const double n = (double)(total_sites - (long long)*active_sites);
    stats[c] += n * v;               // replica 0, sum
    stats[C + c] += n * v * v;       // replica 0, sum of squares
}

// G[v][tap*Cout + c] = dz[do][ix+1-a][iy+1-b][c] for the output site that read voxel v through tap
// (kd,a,b): do*sd - pd + kd = iz; zero where that site does not exist.
__global__ void sparse_conv_gather_dz(const float *__restrict__ dz, const long long *__restrict__ coords, int V,
                                      float *__restrict__ G, SGeom g, FrameMap fm) {
    const int c4n = g.Cout >> 2;
    const long long total = (long long)V * 27 * c4n;
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int ct = (int)(e % c4n);
        const long long r = e / c4n;
        const int tap = (int)(r % 27), v = (int)(r / 27);
        const int kd = tap / 9, a = (tap % 9) / 3, b = tap % 3;
        const long long *cd = coords + (size_t)v * 4;
        const int t = (int)cd[3] + g.pd - kd, y = (int)cd[1] + 1 - a, x = (int)cd[2] + 1 - b;
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t >= 0 && (t % g.sd) == 0 && t / g.sd < g.Dout && y >= 0 && y < g.H && x >= 0 && x < g.W)
            val = *(const float4 *)(dz + (((size_t)(fm_frame_of(fm, v) * g.Dout + t / g.sd) * g.H + y) * g.W + x) * g.Cout + ct * 4);
        *(float4 *)(G + ((size_t)v * 27 + tap) * g.Cout + ct * 4) = val;
    }
}

}  // namespace

extern "C" size_t mvx_index_grid_bytes_frames(int32_t d, int32_t h, int32_t w, int32_t n_frames) {
    if (d <= 0 || h <= 0 || w <= 0 || n_frames <= 0) return 0;
    return sizeof(int32_t) * ((size_t)n_frames * d * h * w + (size_t)n_frames * d * mvx_cdiv(h, OTH) * mvx_cdiv(w, OTW) + 4 + n_frames);
}

extern "C" size_t mvx_index_grid_bytes(int32_t d, int32_t h, int32_t w) { return mvx_index_grid_bytes_frames(d, h, w, 1); }

extern "C" int mvx_index_grid_frames(const int64_t *coords, int32_t n_voxels, int32_t d, int32_t h, int32_t w,
                                     int32_t *grid, int32_t *status, const mvx_frames_t *frames_host, void *stream) {
    MVX_CHECK_ARG(grid && d > 0 && h > 0 && w > 0 && n_voxels >= 0);
    FrameMap fm;
    MVX_CHECK_ARG(mvx_build_frame_map(fm, frames_host, frames_host ? MVX_ROWS_VOXELS : MVX_ROWS_SINGLE, n_voxels, 1.0));
    hipStream_t st = (hipStream_t)stream;
    const size_t planes = (size_t)fm.F * d;
    int32_t *occ = grid + planes * h * w;             // coarse occupancy follows the site grid
    hipError_t e = hipMemsetAsync(grid, 0xFF, sizeof(int32_t) * planes * h * w, st);   // every site = -1
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(occ, 0, sizeof(int32_t) * planes * mvx_cdiv(h, OTH) * mvx_cdiv(w, OTW), st);
    if (e != hipSuccess) return (int)e;
    if (n_voxels == 0) return MVX_OK;
    MVX_CHECK_ARG(coords);
    hipLaunchKernelGGL(index_grid_fill, dim3(mvx_cdiv(n_voxels, 256)), dim3(256), 0, st, (const long long *)coords,
                       n_voxels, d, h, w, grid, occ, status, fm);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" int mvx_index_grid(const int64_t *coords, int32_t n_voxels, int32_t d, int32_t h, int32_t w,
                              int32_t *grid, int32_t *status, void *stream) {
    return mvx_index_grid_frames(coords, n_voxels, d, h, w, grid, status, nullptr, stream);
}

static int sgeom_ok(int32_t din, int32_t dout, int32_t h, int32_t w, int32_t cout, int32_t sd, int32_t pd) {
    if (din <= 0 || dout <= 0 || h <= 0 || w <= 0 || cout <= 0 || cout % 4 || cout > 1024) return 0;
    if (sd < 1 || sd > 2 || pd < 0 || pd > 1) return 0;
    return dout == (din + 2 * pd - 3) / sd + 1;
}

static int sparse_conv_output_impl(const float *p, const int32_t *index_grid, const float *bias, float *out,
                                   double *stats, int32_t din, int32_t dout, int32_t h, int32_t w, int32_t cout,
                                   int32_t stride_d, int32_t pad_d, int32_t flags, int32_t n_frames, const int32_t *tile_flags,
                                   void *stream) {
    const int relu = flags & MVX_FLAG_RELU;
    MVX_CHECK_ARG(index_grid && out && sgeom_ok(din, dout, h, w, cout, stride_d, pad_d));
    MVX_CHECK_ARG(cout / 4 * OTW <= 256 && 256 % (cout / 4) == 0);
    MVX_CHECK_ARG(n_frames >= 1 && n_frames <= MVX_MAX_FRAMES);
    hipStream_t st = (hipStream_t)stream;
    if (stats && !(flags & MVX_FLAG_PREZEROED)) {
        hipError_t e = hipMemsetAsync(stats, 0, sizeof(double) * MVX_REP * 2 * cout * n_frames, st);
        if (e != hipSuccess) return (int)e;
    }
    SGeom g{din, dout, h, w, cout, stride_d, pad_d};
    // the buffer of mvx_index_grid: site grid, coarse occupancy, then one scratch counter per frame
    int32_t *occ = (int32_t *)index_grid + (size_t)n_frames * din * h * w;
    int32_t *active = occ + (size_t)n_frames * din * mvx_cdiv(h, OTH) * mvx_cdiv(w, OTW);
    if (stats) {
        hipError_t e = hipMemsetAsync(active, 0, sizeof(int32_t) * n_frames, st);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(sparse_conv_output, dim3(mvx_cdiv(w, OTW) * mvx_cdiv(h, OTH), dout * n_frames), dim3(256), 0, st, p,
                       index_grid, (const int *)occ, bias, out, stats, g, relu, active, (flags & MVX_FLAG_NO_BG_FILL) ? 1 : 0,
                       (const int *)tile_flags);
    MVX_LAUNCH_CHECK();
    if (stats) {
        hipLaunchKernelGGL(sparse_stats_fix, dim3(mvx_cdiv(cout, 64), n_frames), dim3(64), 0, st, stats, bias,
                           (const int *)active, (long long)dout * h * w, cout, relu);
        MVX_LAUNCH_CHECK();
    }
    return MVX_OK;
}

extern "C" int mvx_sparse_conv_output_frames(const float *p, const int32_t *index_grid, const float *bias, float *out,
                                             double *stats, int32_t din, int32_t dout, int32_t h, int32_t w, int32_t cout,
                                             int32_t stride_d, int32_t pad_d, int32_t flags, int32_t n_frames, void *stream) {
    return sparse_conv_output_impl(p, index_grid, bias, out, stats, din, dout, h, w, cout, stride_d, pad_d, flags, n_frames, nullptr,
                                   stream);
}

// ... with the tile flags of the layer's OUTPUT (mvx_activity_dilate_frames on the index grid: [n_frames * dout][tiles], non-zero =
// the 8 x 16 tile holds a site with a voxel under its taps): only those tiles are built; the others are filled with ReLU(bias)
// or, under MVX_FLAG_NO_BG_FILL, left unwritten.
extern "C" int mvx_sparse_conv_output_tiles_frames(const float *p, const int32_t *index_grid, const float *bias, float *out,
                                                   double *stats, int32_t din, int32_t dout, int32_t h, int32_t w, int32_t cout,
                                                   int32_t stride_d, int32_t pad_d, int32_t flags, int32_t n_frames,
                                                   const int32_t *tile_flags, void *stream) {
    MVX_CHECK_ARG(tile_flags);
    return sparse_conv_output_impl(p, index_grid, bias, out, stats, din, dout, h, w, cout, stride_d, pad_d, flags, n_frames,
                                   tile_flags, stream);
}

extern "C" int mvx_sparse_conv_output(const float *p, const int32_t *index_grid, const float *bias, float *out,
                                      double *stats, int32_t din, int32_t dout, int32_t h, int32_t w, int32_t cout,
                                      int32_t stride_d, int32_t pad_d, int32_t flags, void *stream) {
    return mvx_sparse_conv_output_frames(p, index_grid, bias, out, stats, din, dout, h, w, cout, stride_d, pad_d, flags, 1,
                                         stream);
}

extern "C" int mvx_sparse_conv_gather_dz_frames(const float *dz, const int64_t *coords, int32_t n_voxels, float *g_rows,
                                                int32_t din, int32_t dout, int32_t h, int32_t w, int32_t cout,
                                                int32_t stride_d, int32_t pad_d, const mvx_frames_t *frames_host,
                                                void *stream) {
    MVX_CHECK_ARG(dz && g_rows && n_voxels >= 0 && sgeom_ok(din, dout, h, w, cout, stride_d, pad_d));
    if (n_voxels == 0) return MVX_OK;
    MVX_CHECK_ARG(coords);
    FrameMap fm;
    MVX_CHECK_ARG(mvx_build_frame_map(fm, frames_host, frames_host ? MVX_ROWS_VOXELS : MVX_ROWS_SINGLE, n_voxels, 1.0));
    SGeom g{din, dout, h, w, cout, stride_d, pad_d};
    const long long total = (long long)n_voxels * 27 * (cout / 4);
    hipLaunchKernelGGL(sparse_conv_gather_dz, dim3(mvx_cdiv(total, 256) > 4096 ? 4096 : mvx_cdiv(total, 256)), dim3(256),
                       0, (hipStream_t)stream, dz, (const long long *)coords, n_voxels, g_rows, g, fm);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" int mvx_sparse_conv_gather_dz(const float *dz, const int64_t *coords, int32_t n_voxels, float *g_rows,
                                         int32_t din, int32_t dout, int32_t h, int32_t w, int32_t cout,
                                         int32_t stride_d, int32_t pad_d, void *stream) {
    return mvx_sparse_conv_gather_dz_frames(dz, coords, n_voxels, g_rows, din, dout, h, w, cout, stride_d, pad_d, nullptr,
                                            stream);
}
