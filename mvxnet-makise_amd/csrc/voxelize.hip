// Point-cloud voxelizer for gfx950.
//
// Replaces the reference's sequential hash-map grouping (cpp/voxelutil.cpp:325-360 and the
// Python loop of modules/data/Preprocessing.py:94-116) with five stream-ordered launches that
// reproduce its ORDER semantics exactly:
//   voxel order  = order of first appearance in the (shuffled) point stream,
//   kept points  = the first T stream positions that fall in the voxel, in stream order,
//   index math   = (int32)(((double)xyz - low) / size) in f64 with true division,
//   centroid     = sequential f64 (or f32, 7-channel mode) sum in stream order / count.
//
//   K0 init     : hash table reset
//   K1 insert   : one thread per stream position -> open-addressing insert keyed by the packed
//                 (ix,iy,iz); atomicMin records the first stream position of each key
//   K2 scan     : one 1024-thread workgroup per frame: first-appearance flags -> voxel ids,
//                 then per-voxel point counts -> segment offsets (CSR)
//   K3 append   : every stream position joins its voxel's segment (unordered)
//   K4 gather   : one wave per voxel: selects the T smallest stream positions of the segment in
//                 order (wave-wide rank counting), stages the point group in LDS, reduces the
//                 centroid, and writes the [T][C] block with coalesced stores.
#include "common.h"

namespace {

constexpr unsigned long long KEY_EMPTY = ~0ull;
constexpr int KEY_BIAS = 1 << 20;

struct VoxWs {
    unsigned long long *keys;  // [F][slots]
    int *first;                // [F][slots] smallest stream position of the key
    int *scount;               // [F][slots] points with this key
    int *slot_vid;             // [F][slots] voxel id of the key
    int *slot_of;              // [F][cap]   slot of stream position s
    int *vox_slot;             // [F][cap]   slot of voxel v
    int *seg_off;              // [F][cap+1] CSR offsets of voxel segments
    int *cursor;               // [F][cap]
    int *members;              // [F][cap]   stream positions grouped by voxel
    int slots;
};

__host__ inline int table_slots(int cap_points) {
    int s = 64;
    while (s < 2 * cap_points) s <<= 1;
    return s;
}

__host__ inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

__host__ VoxWs carve(void *ws, int F, int cap, size_t *total) {
    VoxWs w;
    w.slots = table_slots(cap);
    size_t off = 0;
    char *base = (char *)ws;
    auto take = [&](size_t bytes) { char *p = base ? base + off : nullptr; off += align256(bytes); return p; };
    w.keys = (unsigned long long *)take((size_t)F * w.slots * 8);
    w.first = (int *)take((size_t)F * w.slots * 4);
    w.scount = (int *)take((size_t)F * w.slots * 4);
    w.slot_vid = (int *)take((size_t)F * w.slots * 4);
    w.slot_of = (int *)take((size_t)F * cap * 4);
    w.vox_slot = (int *)take((size_t)F * cap * 4);
    w.seg_off = (int *)take((size_t)F * (cap + 1) * 4);
    w.cursor = (int *)take((size_t)F * cap * 4);
    w.members = (int *)take((size_t)F * cap * 4);
    *total = off;
    return w;
}

__global__ void vox_init(VoxWs w, int cap) {
    const int f = blockIdx.y;
    const int stride = gridDim.x * blockDim.x;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < w.slots; i += stride) {
        w.keys[(size_t)f * w.slots + i] = KEY_EMPTY;
        w.first[(size_t)f * w.slots + i] = 0x7fffffff;
        w.scount[(size_t)f * w.slots + i] = 0;
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += stride)
        w.cursor[(size_t)f * cap + i] = 0;
}

__device__ __forceinline__ unsigned hash_key(unsigned long long k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33;
    return (unsigned)k;
}

__global__ void vox_insert(const float *__restrict__ pcd, const int *__restrict__ perm,
                           const int *__restrict__ n_points, const int *__restrict__ ext_idx,
                           int cap, int ncol,
                           double lx, double ly, double lz, double sx, double sy, double sz,
                           VoxWs w, int *status) {
    const int f = blockIdx.y;
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = min(n_points[f], cap);
    if (s >= n) return;
    const int p = perm ? perm[(size_t)f * cap + s] : s;
    const float *row = pcd + ((size_t)f * cap + p) * ncol;
    // Preprocessing.py:87-90 -- f64 subtract, f64 true division, truncation toward zero
    int ix, iy, iz;
    if (ext_idx) {
        const int *e = ext_idx + ((size_t)f * cap + s) * 3;
        ix = e[0]; iy = e[1]; iz = e[2];
    } else {
        ix = (int)(((double)row[0] - lx) / sx);
        iy = (int)(((double)row[1] - ly) / sy);
        iz = (int)(((double)row[2] - lz) / sz);
    }
    const unsigned ux = (unsigned)(ix + KEY_BIAS), uy = (unsigned)(iy + KEY_BIAS), uz = (unsigned)(iz + KEY_BIAS);
    if ((ux | uy | uz) >> 21) atomicOr(status, 1);
    const unsigned long long key = (unsigned long long)(ux & 0x1fffff) |
                                   ((unsigned long long)(uy & 0x1fffff) << 21) |
                                   ((unsigned long long)(uz & 0x1fffff) << 42);
    unsigned long long *keys = w.keys + (size_t)f * w.slots;
    const unsigned mask = (unsigned)w.slots - 1;
    unsigned h = hash_key(key) & mask;
    while (true) {
        unsigned long long cur = keys[h];
        if (cur == key) break;
        if (cur == KEY_EMPTY) {
            unsigned long long old = atomicCAS(&keys[h], KEY_EMPTY, key);
            if (old == KEY_EMPTY || old == key) break;
        }
        h = (h + 1) & mask;
    }
    atomicMin(&w.first[(size_t)f * w.slots + h], s);
    atomicAdd(&w.scount[(size_t)f * w.slots + h], 1);
    w.slot_of[(size_t)f * cap + s] = (int)h;
}

// One workgroup per frame.  Each thread owns a CONTIGUOUS run of stream positions (then of voxel ids), issues all its
// dependent loads up front, scans its run locally and takes part in ONE block scan per phase -- the previous form
// walked the stream in 1024-element rounds with a block scan (three barriers and two dependent loads) per round:
// 25 serial rounds = 50 us per call whatever the batch size.
constexpr int SCAN_K = 8;                 // positions per thread per pass (1024 x 8 = 8192 positions per pass)
__global__ __launch_bounds__(1024) void vox_scan(const int *__restrict__ n_points, int cap, int cap_voxels,
                                                 VoxWs w, int *n_voxels, int *status) {
    __shared__ int smem[17];
    const int f = blockIdx.x;
    const int n = min(n_points[f], cap);
    const int *first = w.first + (size_t)f * w.slots;
    const int *scount = w.scount + (size_t)f * w.slots;
    const int *slot_of = w.slot_of + (size_t)f * cap;
    int *slot_vid = w.slot_vid + (size_t)f * w.slots;
    int *vox_slot = w.vox_slot + (size_t)f * cap;
    int *seg_off = w.seg_off + (size_t)f * (cap + 1);
    int base = 0;
    for (int t0 = 0; t0 < n; t0 += blockDim.x * SCAN_K) {
        const int s0 = t0 + threadIdx.x * SCAN_K;
        int slot[SCAN_K], flag[SCAN_K], mine = 0;
#pragma unroll
        for (int j = 0; j < SCAN_K; ++j) slot[j] = s0 + j < n ? slot_of[s0 + j] : -1;
#pragma unroll
        for (int j = 0; j < SCAN_K; ++j) {
            flag[j] = slot[j] >= 0 && first[slot[j]] == s0 + j;       // "I am my voxel's first point"
            mine += flag[j];
        }
        int tot;
        int v = base + block_excl_scan_i32(mine, smem, &tot);
#pragma unroll
        for (int j = 0; j < SCAN_K; ++j)
            if (flag[j]) {
                slot_vid[slot[j]] = v;
                vox_slot[v] = slot[j];
                ++v;
            }
        base += tot;
    }
    const int V = base;
    if (threadIdx.x == 0) {
        n_voxels[f] = V;
        if (V > cap_voxels) atomicOr(status, 2);
    }
    __syncthreads();   // vox_slot[] written above is read below by other threads of this block
    int run = 0;
    for (int t0 = 0; t0 < V; t0 += blockDim.x * SCAN_K) {
        const int v0 = t0 + threadIdx.x * SCAN_K;
        int c[SCAN_K], mine = 0;
#pragma unroll
        for (int j = 0; j < SCAN_K; ++j) c[j] = v0 + j < V ? vox_slot[v0 + j] : -1;
#pragma unroll
        for (int j = 0; j < SCAN_K; ++j) {
            c[j] = c[j] >= 0 ? scount[c[j]] : 0;
            mine += c[j];
        }
        int tot;
        int off = run + block_excl_scan_i32(mine, smem, &tot);
#pragma unroll
        for (int j = 0; j < SCAN_K; ++j)
            if (v0 + j < V) {
                seg_off[v0 + j] = off;
                off += c[j];
            }
        run += tot;
    }
    if (threadIdx.x == 0) seg_off[V] = run;
}

__global__ void vox_append(const int *__restrict__ n_points, int cap, VoxWs w) {
    const int f = blockIdx.y;
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = min(n_points[f], cap);
    if (s >= n) return;
    const int v = w.slot_vid[(size_t)f * w.slots + w.slot_of[(size_t)f * cap + s]];
    const int pos = atomicAdd(&w.cursor[(size_t)f * cap + v], 1);
    w.members[(size_t)f * cap + w.seg_off[(size_t)f * (cap + 1) + v] + pos] = s;
}

// One wave per voxel.  LDS per wave: 64 candidate slots + the staged point group.
template <int C>
__global__ __launch_bounds__(256) void vox_gather(const float *__restrict__ pcd, const int *__restrict__ perm,
                                                  int cap, int ncol, int T, int cap_voxels, VoxWs w,
                                                  const int *__restrict__ n_voxels,
                                                  float *__restrict__ voxels, long long *__restrict__ coords,
                                                  int *__restrict__ counts) {
    __shared__ int s_best[4][64];
    __shared__ float s_pts[4][64][6];
    const int f = blockIdx.y;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int v = blockIdx.x * 4 + wid;
    const int V = min(n_voxels[f], cap_voxels);
    if (v >= V) return;   // whole wave leaves together: no block-wide barrier below
    const int *seg_off = w.seg_off + (size_t)f * (cap + 1);
    const int *members = w.members + (size_t)f * cap;
    const int beg = seg_off[v], n = seg_off[v + 1] - beg;

    // ---- the (up to) 64 smallest stream positions of the segment, sorted: lane r holds rank r
    int best = 0x7fffffff;
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int cand = (c0 + lane < n) ? members[beg + c0 + lane] : 0x7fffffff;
        int rb = 0, rc = 0;   // ranks of `best` and `cand` inside best U cand (positions are unique)
        for (int j = 0; j < 64; ++j) {
            const int bj = __shfl(best, j, 64), cj = __shfl(cand, j, 64);
            rb += (bj < best) + (cj < best);
            rc += (bj < cand) + (cj < cand);
        }
        // sentinels tie with each other; give them distinct ranks past every real value
        if (best == 0x7fffffff) rb = 128 + lane;
        if (cand == 0x7fffffff) rc = 192 + lane;
        __builtin_amdgcn_wave_barrier();
        s_best[wid][lane] = 0x7fffffff;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (rb < 64) s_best[wid][rb] = best;
        if (rc < 64) s_best[wid][rc] = cand;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        best = s_best[wid][lane];
    }
    const int kept = min(n, T);

    // ---- stage the kept point group in LDS
    float px = 0.f, py = 0.f, pz = 0.f;
    if (lane < kept) {
        const int p = perm ? perm[(size_t)f * cap + best] : best;
        const float *row = pcd + ((size_t)f * cap + p) * ncol;
        px = row[0]; py = row[1]; pz = row[2];
        s_pts[wid][lane][0] = px;
        s_pts[wid][lane][1] = py;
        s_pts[wid][lane][2] = pz;
        s_pts[wid][lane][3] = row[3];
        s_pts[wid][lane][4] = ncol > 4 ? row[4] : 0.f;
        s_pts[wid][lane][5] = ncol > 5 ? row[5] : 0.f;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- centroid: sequential sum in stream order (Preprocessing.py:112-113 / :71)
    double cx, cy, cz;
    if (C == 9) {
        double sx = 0.0, sy = 0.0, sz = 0.0;
        for (int j = 0; j < kept; ++j) {
            sx += (double)s_pts[wid][j][0];
            sy += (double)s_pts[wid][j][1];
            sz += (double)s_pts[wid][j][2];
        }
        cx = sx / (double)kept; cy = sy / (double)kept; cz = sz / (double)kept;
    } else {
        float sx = 0.f, sy = 0.f, sz = 0.f;
        for (int j = 0; j < kept; ++j) {
            sx += s_pts[wid][j][0];
            sy += s_pts[wid][j][1];
            sz += s_pts[wid][j][2];
        }
        cx = (double)sx / (double)kept; cy = (double)sy / (double)kept; cz = (double)sz / (double)kept;
    }

    // ---- coalesced write of the [T][C] block; padded rows carry -centroid in cols 3:6
    //      (Preprocessing.py:115)
    float *out = voxels + ((size_t)f * cap_voxels + v) * (size_t)T * C;
    for (int e = lane; e < T * C; e += 64) {
        const int t = e / C, c = e - t * C;
        const bool real = t < kept;
        float val;
        if (c < 3) {
            val = real ? s_pts[wid][t][c] : 0.f;
        } else if (c < 6) {
            const float x = real ? s_pts[wid][t][c - 3] : 0.f;
            const double cen = c == 3 ? cx : (c == 4 ? cy : cz);
            val = (float)((double)x - cen);
        } else {
            val = real ? s_pts[wid][t][c - 3] : 0.f;
        }
        out[e] = val;
    }
    if (lane == 0) {
        const unsigned long long key = w.keys[(size_t)f * w.slots + w.vox_slot[(size_t)f * cap + v]];
        long long *cd = coords + ((size_t)f * cap_voxels + v) * 4;
        cd[0] = 0;
        cd[1] = (long long)(int)(key & 0x1fffff) - KEY_BIAS;
        cd[2] = (long long)(int)((key >> 21) & 0x1fffff) - KEY_BIAS;
        cd[3] = (long long)(int)((key >> 42) & 0x1fffff) - KEY_BIAS;
        counts[(size_t)f * cap_voxels + v] = kept;
    }
}

}  // namespace

extern "C" int mvx_abi_version(void) { return 2; }

extern "C" size_t mvx_voxelize_workspace_bytes(int32_t n_frames, int32_t cap_points) {
    if (n_frames <= 0 || cap_points <= 0) return 0;
    size_t total = 0;
    carve(nullptr, n_frames, cap_points, &total);
    return total;
}

extern "C" int mvx_voxelize(const float *pcd, const int32_t *perm, const int32_t *n_points,
                            const int32_t *ext_idx, int32_t n_frames, int32_t cap_points, int32_t ncol,
                            double lo_x, double lo_y, double lo_z,
                            double size_x, double size_y, double size_z,
                            int32_t T, int32_t out_channels, int32_t cap_voxels,
                            float *voxels, int64_t *coords, int32_t *counts, int32_t *n_voxels,
                            int32_t *status, void *workspace, size_t workspace_bytes, void *stream) {
    MVX_CHECK_ARG(pcd && n_points && voxels && coords && counts && n_voxels && status && workspace);
    MVX_CHECK_ARG(n_frames > 0 && cap_points > 0 && cap_voxels > 0 && ncol >= 4);
    MVX_CHECK_ARG(T > 0 && T <= 64);
    MVX_CHECK_ARG(out_channels == 7 || out_channels == 9);
    MVX_CHECK_ARG(size_x > 0 && size_y > 0 && size_z > 0);
    if ((long long)cap_points > (1ll << 28)) return MVX_ESIZE;
    size_t need = 0;
    VoxWs w = carve(workspace, n_frames, cap_points, &need);
    MVX_CHECK_ARG(workspace_bytes >= need);
    hipStream_t st = (hipStream_t)stream;

    const unsigned gb = mvx_cdiv(cap_points, 256);
    hipLaunchKernelGGL(vox_init, dim3(mvx_cdiv(w.slots, 256), n_frames), dim3(256), 0, st, w, cap_points);
    MVX_LAUNCH_CHECK();
    hipLaunchKernelGGL(vox_insert, dim3(gb, n_frames), dim3(256), 0, st, pcd, perm, n_points, ext_idx, cap_points, ncol,
                       lo_x, lo_y, lo_z, size_x, size_y, size_z, w, status);
    MVX_LAUNCH_CHECK();
    hipLaunchKernelGGL(vox_scan, dim3(n_frames), dim3(1024), 0, st, n_points, cap_points, cap_voxels, w,
                       n_voxels, status);
    MVX_LAUNCH_CHECK();
    hipLaunchKernelGGL(vox_append, dim3(gb, n_frames), dim3(256), 0, st, n_points, cap_points, w);
    MVX_LAUNCH_CHECK();
    const dim3 gg(mvx_cdiv(cap_voxels < cap_points ? cap_voxels : cap_points, 4), n_frames);
    if (out_channels == 9)
        hipLaunchKernelGGL(vox_gather<9>, gg, dim3(256), 0, st, pcd, perm, cap_points, ncol, T, cap_voxels, w,
                           n_voxels, voxels, (long long *)coords, counts);
    else
        hipLaunchKernelGGL(vox_gather<7>, gg, dim3(256), 0, st, pcd, perm, cap_points, ncol, T, cap_voxels, w,
                           n_voxels, voxels, (long long *)coords, counts);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}
