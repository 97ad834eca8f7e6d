// Point-cloud voxelizer for gfx950.
//
// Replaces the reference's sequential hash-map grouping (cpp/voxelutil.cpp:325-360 and the
// Python loop of modules/data/Preprocessing.py:94-116) with THREE stream-ordered launches (+ two memsets) that
// reproduce its ORDER semantics exactly:
//   voxel order  = order of first appearance in the (shuffled) point stream,
//   kept points  = the first T stream positions that fall in the voxel, in stream order,
//   index math   = (int32)(((double)xyz - low) / size) in f64 with true division,
//   centroid     = sequential f64 (or f32, 7-channel mode) sum in stream order / count.
//
//   K1 insert : one thread per stream position -> open-addressing insert keyed by the packed (ix,iy,iz); atomicMin
//               records the first stream position of each key; the position also joins the key's 64-entry bucket
//               (arrival order, i.e. unordered)
//   K2 scan   : "I am my voxel's first point" flags -> voxel ids in first-appearance order: ONE decoupled-look-back scan
//               over the positions of ALL frames (work-item ticket per block, packed flag|value words, no fences), so a
//               frame's voxels follow the previous frame's; per-frame voxel offsets fall out of it
//   K3 gather : one wave per voxel: sorts the bucket (<= 64 members: wave-wide rank counting over the actual member
//               count) or, for the few voxels with more members, walks the frame's stream in order until T members are
//               found; stages the point group in LDS, reduces the centroid, writes the [T][C] block with coalesced stores.
// Outputs either with a fixed stride per frame ([F][cap_voxels]...) or back to back over all frames (concat: the batch
// layout of the frame-set path, coords[:,0] = frame index = the batch column of train.py:119).
#include <atomic>
#include "common.h"

namespace {

constexpr unsigned long long KEY_EMPTY = ~0ull;
constexpr int KEY_BIAS = 1 << 20;

constexpr int BK = 64;                  // bucket entries per key

struct VoxWs {
    unsigned long long *keys;  // [F][slots]        (memset 0xFF = KEY_EMPTY)
    unsigned *first;           // [F][slots]        smallest stream position of the key (memset 0xFF)
    int *scount;               // [F][slots]        points with this key            (memset 0 from here ...)
    unsigned *ticket;          // [4]               work-item counter of the scan
    unsigned long long *bstate;// [nblk]            look-back words: flag << 62 | value   (... to here)
    int *frame_base;           // [F+1]             id of the first voxel of frame f (written by the scan)
    int *slot_of;              // [F][cap]          slot of stream position s
    int *vox_slot;             // [F*cap]           slot of (global) voxel v
    int *bucket;               // [F][slots][BK]    stream positions of the key, arrival order
    int *wide_list;            // [F*cap]           voxels left to vox_gather_wide (more than 16 members); count in ticket[1]
    int slots, nblk;
    size_t ff_bytes, zero_bytes;   // extents of the two memsets (from keys / from scount)
};

constexpr int SCAN_POS = 1024;          // stream positions per scan workgroup (256 threads x 4)

__host__ inline int table_slots(int cap_points) {
    int s = 64;
    while (s < 2 * cap_points) s <<= 1;
    return s;
}

__host__ inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

__host__ VoxWs carve(void *ws, int F, int cap, size_t *total) {
    VoxWs w;
    w.slots = table_slots(cap);
    w.nblk = (int)(((long long)F * cap + SCAN_POS - 1) / SCAN_POS);
    size_t off = 0;
    char *base = (char *)ws;
    auto take = [&](size_t bytes) { char *p = base ? base + off : nullptr; off += align256(bytes); return p; };
    w.keys = (unsigned long long *)take((size_t)F * w.slots * 8);
    w.first = (unsigned *)take((size_t)F * w.slots * 4);
    w.ff_bytes = off;
    const size_t z0 = off;
    w.scount = (int *)take((size_t)F * w.slots * 4);
    w.ticket = (unsigned *)take(16);
    w.bstate = (unsigned long long *)take((size_t)w.nblk * 8);
    w.zero_bytes = off - z0;
    w.frame_base = (int *)take((size_t)(F + 1) * 4);
    w.slot_of = (int *)take((size_t)F * cap * 4);
    w.vox_slot = (int *)take((size_t)F * cap * 4);
    w.bucket = (int *)take((size_t)F * w.slots * BK * 4);
    w.wide_list = (int *)take((size_t)F * cap * 4);
    *total = off;
    return w;
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
    const size_t slot = (size_t)f * w.slots + h;
    atomicMin(&w.first[slot], (unsigned)s);
    const int t = atomicAdd(&w.scount[slot], 1);
    if (t < BK) w.bucket[slot * BK + t] = s;
    w.slot_of[(size_t)f * cap + s] = (int)h;
}

// Decoupled look-back scan of the first-point flags over the positions of all frames ([F][cap], positions past a
// frame's live count are dead).  A workgroup takes the next 1024 positions from a ticket counter, so a workgroup only
// ever waits for workgroups that started before it (no deadlock whatever the dispatch order).  Look-back words pack
// (flag, value) into one 64-bit atomic: 1 = aggregate of this block, 2 = inclusive prefix up to and including it.
__global__ __launch_bounds__(256) void vox_scan(const int *__restrict__ n_points, int cap, int F, VoxWs w) {
    __shared__ int smem[17];
    __shared__ unsigned s_ticket;
    __shared__ int s_prefix;
    if (threadIdx.x == 0) s_ticket = atomicAdd(w.ticket, 1u);
    __syncthreads();
    const int b = (int)s_ticket;
    const long long total_pos = (long long)F * cap;
    const long long g0 = (long long)b * SCAN_POS + threadIdx.x * 4;
    int slot[4], flag[4], fr[4], mine = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long long g = g0 + j;
        slot[j] = -1; flag[j] = 0; fr[j] = 0;
        if (g < total_pos) {
            const int f = (int)(g / cap), s = (int)(g - (long long)f * cap);
            fr[j] = f;
            if (s < min(n_points[f], cap)) {
                slot[j] = w.slot_of[g];
                flag[j] = w.first[(size_t)f * w.slots + slot[j]] == (unsigned)s;       // "I am my voxel's first point"
            }
        }
        mine += flag[j];
    }
    int tot;
    const int my_off = block_excl_scan_i32(mine, smem, &tot);
    if (threadIdx.x == 0) {
        const unsigned long long word = ((b == 0 ? 2ull : 1ull) << 62) | (unsigned long long)(unsigned)tot;
        __hip_atomic_store(w.bstate + b, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x < 64) {            // wave 0 looks back over the predecessors, 64 at a time
        int prefix = 0;
        int k = b - 1;
        while (k >= 0) {
            const int idx = k - (int)threadIdx.x;
            unsigned long long word = 2ull << 62;           // lanes before block 0: an (empty) inclusive prefix
            if (idx >= 0) {
                do {
                    word = __hip_atomic_load(w.bstate + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } while ((word >> 62) == 0);
            }
            const unsigned long long incl = __ballot((word >> 62) == 2ull);
            const int stop = __ffsll((long long)incl) - 1;                   // nearest predecessor with an inclusive prefix
            const int val = (int)(unsigned)(word & 0xffffffffull);
            int part = ((int)threadIdx.x <= stop || stop < 0) ? val : 0;
            if (idx < 0) part = 0;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
            prefix += part;
            if (stop >= 0) break;
            k -= 64;
        }
        if (threadIdx.x == 0) {
            s_prefix = prefix;
            if (b > 0)
                __hip_atomic_store(w.bstate + b, (2ull << 62) | (unsigned long long)(unsigned)(prefix + tot), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    int v = s_prefix + my_off;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long long g = g0 + j;
        if (g < total_pos && g - (long long)fr[j] * cap == 0) w.frame_base[fr[j]] = v;     // first position of a frame
        if (flag[j]) w.vox_slot[v++] = slot[j];
    }
    if (g0 <= total_pos - 1 && total_pos - 1 < g0 + 4) w.frame_base[F] = v;               // the thread that owns the last position
}

// Four voxels per wave.  A lidar voxel holds 3.9 points on average, so a wave per voxel (the first form of this kernel) ran its
// five dependent loads -- voxel -> slot -> member count -> bucket -> permutation -> point row -- with four live lanes: 0.025 of the
// HBM roofline.  Here a 16-lane group owns a voxel with up to 16 members: it ranks the bucket with 16 shuffles inside the group,
// stages its points in LDS, sums the centroid in stream order and writes the [T][C] block 64 bytes at a time, four voxels side by
// side.  A voxel with more members (or a crowded one whose bucket overflowed) is handled afterwards by the whole wave
// (gather_wide: the wave-per-voxel form).  Same outputs bit for bit: member order, centroid summation order and index math are
// those of cpp/voxelutil.cpp:325-360 / Preprocessing.py:94-116 (see the file header).
template <int C>
__device__ __forceinline__ void gather_wide(const float *__restrict__ pcd, const int *__restrict__ perm,
                                            const int *__restrict__ n_points, int cap, int ncol, int T, const VoxWs &w,
                                            float *__restrict__ voxels, long long *__restrict__ coords, int *__restrict__ counts,
                                            int f, long long dst, int h, int n, int concat, int (*s_best)[64], float (*s_pts)[64][6],
                                            int wid, int lane) {
    const size_t slot = (size_t)f * w.slots + h;
    // ---- the (up to) T smallest stream positions of the voxel, sorted: lane r holds rank r
    int best = 0x7fffffff;
    if (n <= BK) {
        const int cand = lane < n ? w.bucket[slot * BK + lane] : 0x7fffffff;
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += __shfl(cand, j, 64) < cand;       // positions are unique
        __builtin_amdgcn_wave_barrier();
        s_best[wid][lane] = 0x7fffffff;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane < n) s_best[wid][rank] = cand;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        best = s_best[wid][lane];
    } else {
        // crowded voxel: the bucket holds an arbitrary subset -- walk the frame's stream in order, the first T matches
        // are the answer (already sorted)
        const int npts = min(n_points[f], cap);
        const int *so = w.slot_of + (size_t)f * cap;
        // 1,024 positions per round, their sixteen loads in flight together: one load per round made the walk a chain of ~170
        // memory latencies (0.1 ms for ONE such voxel -- the whole kernel's duration, whatever the other 80,000 cost)
        constexpr int WALK = 16;
        int found = 0;
        for (int s0 = 0; s0 < npts && found < T; s0 += 64 * WALK) {
            int sv[WALK];
#pragma unroll
            for (int u = 0; u < WALK; ++u) {
                const int s = s0 + u * 64 + lane;
                sv[u] = s < npts ? so[s] : -1;
            }
#pragma unroll
            for (int u = 0; u < WALK; ++u) {
                const bool match = sv[u] == h;
                const unsigned long long bal = __ballot(match);
                const int pos = found + __popcll(bal & ((1ull << lane) - 1ull));
                if (match && pos < 64) s_best[wid][pos] = s0 + u * 64 + lane;
                found += __popcll(bal);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        best = lane < min(found, 64) ? s_best[wid][lane] : 0x7fffffff;
    }
    const int kept = min(n, T);

    // ---- stage the kept point group in LDS
    if (lane < kept) {
        const int p = perm ? perm[(size_t)f * cap + best] : best;
        const float *row = pcd + ((size_t)f * cap + p) * ncol;
        s_pts[wid][lane][0] = row[0];
        s_pts[wid][lane][1] = row[1];
        s_pts[wid][lane][2] = row[2];
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

    // ---- coalesced write of the [T][C] block; padded rows carry -centroid in cols 3:6 (Preprocessing.py:115)
    float *out = voxels + (size_t)dst * (size_t)T * C;
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
        const unsigned long long key = w.keys[slot];
        long long *cd = coords + (size_t)dst * 4;
        cd[0] = concat ? f : 0;                       // batch column (train.py:119): the frame inside a concatenated batch
        cd[1] = (long long)(int)(key & 0x1fffff) - KEY_BIAS;
        cd[2] = (long long)(int)((key >> 21) & 0x1fffff) - KEY_BIAS;
        cd[3] = (long long)(int)((key >> 42) & 0x1fffff) - KEY_BIAS;
        counts[dst] = kept;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();                   // s_best / s_pts are reused by the next wide voxel of this wave
}

constexpr int VG = 16;                  // lanes per voxel in the narrow path

template <int C>
__global__ __launch_bounds__(256) void vox_gather(const float *__restrict__ pcd, const int *__restrict__ perm,
                                                  const int *__restrict__ n_points, int cap, int ncol, int T, int F,
                                                  int cap_voxels, int concat, VoxWs w, float *__restrict__ voxels,
                                                  long long *__restrict__ coords, int *__restrict__ counts,
                                                  int *__restrict__ n_voxels, int *__restrict__ vox_off,
                                                  int *__restrict__ status) {
    extern __shared__ float s_img[];                   // [wave][group][T * C] (sized by the launch: 16 * T * C floats)
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int sub = lane >> 4, sl = lane & (VG - 1);
    if (blockIdx.x == 0 && threadIdx.x <= F) {          // per-frame counts / offsets for the caller
        const int f = threadIdx.x;
        if (vox_off) vox_off[f] = w.frame_base[f];
        if (f < F) {
            const int nv = w.frame_base[f + 1] - w.frame_base[f];
            n_voxels[f] = nv;
            if (!concat && nv > cap_voxels) atomicOr(status, 2);
        }
        if (f == F && concat && w.frame_base[F] > cap_voxels) atomicOr(status, 2);
    }
    // frame of the voxels: the F + 1 bases in ONE load (lane k holds base k; F < 64)
    const int fb = lane <= F ? w.frame_base[lane] : 0x7fffffff;
    const int Vtot = __shfl(fb, F, 64);
    // the launch cannot know the voxel count (it lives on the device): a fixed grid walks the voxels, so that the cost follows the
    // voxels that exist and not the capacity (the first form launched a wave per POSSIBLE voxel: 480 k waves that loaded the
    // bases and left -- 59 rounds of the chip's wave slots -- were most of its 0.12 ms)
    // neighbouring voxel ids go to DIFFERENT waves (group g of wave W takes voxel base + g * waves + W): ids follow first
    // appearance in the stream, so a frame's first ids are its most populated voxels -- the ones that need the wide path, which
    // a wave runs one after the other
    const int nw = gridDim.x * 4, wv = blockIdx.x * 4 + wid;
    for (int base = 0; base < Vtot; base += nw * 4) {   // whole waves: no block-wide barrier below
    const int v = base + sub * nw + wv;
    int f = 0;
    for (int k = 1; k < F; ++k) f += v >= __shfl(fb, k, 64);
    f = v < Vtot ? f : 0;
    const int vl = v - __shfl(fb, f, 64);
    const long long dst = concat ? (long long)v : (long long)f * cap_voxels + vl;
    const bool ok = v < Vtot && (concat ? v < cap_voxels : vl < cap_voxels);
    const int h = ok ? w.vox_slot[v] : 0;
    const size_t slot = (size_t)f * w.slots + h;
    const int n = ok ? w.scount[slot] : 0;
    const bool narrow = ok && n <= VG && n <= T;

    if (narrow) {
        // ---- rank the (at most 16) members inside the group; padding candidates are distinct and larger than any position
        const int cand = sl < n ? w.bucket[slot * BK + sl] : 0x7fffff00 + sl;
        int rank = 0;
#pragma unroll
        for (int j = 0; j < VG; ++j) rank += __shfl(cand, (lane & ~(VG - 1)) + j, 64) < cand;
        // lane of rank r receives the r-th smallest position (a permutation inside the group)
        const int best = __builtin_amdgcn_ds_permute(((lane & ~(VG - 1)) + rank) << 2, cand);
        const int kept = n;
        float px = 0.f, py = 0.f, pz = 0.f, p3 = 0.f, p4 = 0.f, p5 = 0.f;
        if (sl < kept) {
            const int p = perm ? perm[(size_t)f * cap + best] : best;
            const float *row = pcd + ((size_t)f * cap + p) * ncol;
            px = row[0]; py = row[1]; pz = row[2]; p3 = row[3];
            p4 = ncol > 4 ? row[4] : 0.f;
            p5 = ncol > 5 ? row[5] : 0.f;
        }
        // ---- centroid: sequential sum in stream order over the group's lanes
        double cx, cy, cz;
        const int g0 = lane & ~(VG - 1);
        if (C == 9) {
            double sx = 0.0, sy = 0.0, sz = 0.0;
#pragma unroll 2
            for (int j = 0; j < VG; ++j) {
                const float xj = __shfl(px, g0 + j, 64), yj = __shfl(py, g0 + j, 64), zj = __shfl(pz, g0 + j, 64);
                if (j < kept) { sx += (double)xj; sy += (double)yj; sz += (double)zj; }
            }
            cx = sx / (double)kept; cy = sy / (double)kept; cz = sz / (double)kept;
        } else {
            float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll 2
            for (int j = 0; j < VG; ++j) {
                const float xj = __shfl(px, g0 + j, 64), yj = __shfl(py, g0 + j, 64), zj = __shfl(pz, g0 + j, 64);
                if (j < kept) { sx += xj; sy += yj; sz += zj; }
            }
            cx = (double)sx / (double)kept; cy = (double)sy / (double)kept; cz = (double)sz / (double)kept;
        }
        // ---- the [T][C] block is laid out in LDS -- lane t forms the nine numbers of point t ONCE (the f64 subtraction of the
        // centroid per point, not per output element), the lanes share the padded rows (which carry -centroid in cols 3:6,
        // Preprocessing.py:115) -- and then copied out, 64 bytes per group and store, without any index arithmetic
        float *img = s_img + (size_t)(wid * 4 + sub) * T * C;
        const float ncx = (float)(0.0 - cx), ncy = (float)(0.0 - cy), ncz = (float)(0.0 - cz);
        for (int t = sl; t < T; t += VG) {
            float *r = img + t * C;
            if (t < kept) {                            // t == sl: this lane's own point
                r[0] = px; r[1] = py; r[2] = pz;
                r[3] = (float)((double)px - cx); r[4] = (float)((double)py - cy); r[5] = (float)((double)pz - cz);
                r[6] = p3;
                if (C == 9) { r[7] = p4; r[8] = p5; }
            } else {
                r[0] = 0.f; r[1] = 0.f; r[2] = 0.f; r[3] = ncx; r[4] = ncy; r[5] = ncz; r[6] = 0.f;
                if (C == 9) { r[7] = 0.f; r[8] = 0.f; }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float *out = voxels + (size_t)dst * (size_t)T * C;
        for (int e = sl; e < T * C; e += VG) out[e] = img[e];
        if (sl == 0) {
            const unsigned long long key = w.keys[slot];
            long long *cd = coords + (size_t)dst * 4;
            cd[0] = concat ? f : 0;
            cd[1] = (long long)(int)(key & 0x1fffff) - KEY_BIAS;
            cd[2] = (long long)(int)((key >> 21) & 0x1fffff) - KEY_BIAS;
            cd[3] = (long long)(int)((key >> 42) & 0x1fffff) - KEY_BIAS;
            counts[dst] = kept;
        }
    }
    // ---- a voxel that does not fit a group is left to vox_gather_wide (a wave per voxel): listed, in any order
    if (ok && !narrow && sl == 0) w.wide_list[atomicAdd(&w.ticket[1], 1u)] = v;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();                   // the next round reuses the wave's LDS
    }
}

// The voxels vox_gather listed (more than 16 members, 1.5 % on lidar frames): a wave per voxel, a fixed grid over the list
template <int C>
__global__ __launch_bounds__(256) void vox_gather_wide(const float *__restrict__ pcd, const int *__restrict__ perm,
                                                       const int *__restrict__ n_points, int cap, int ncol, int T, int F,
                                                       int cap_voxels, int concat, VoxWs w, float *__restrict__ voxels,
                                                       long long *__restrict__ coords, int *__restrict__ counts) {
    __shared__ int s_best[4][64];
    __shared__ float s_pts[4][64][6];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int count = (int)__hip_atomic_load(&w.ticket[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int fb = lane <= F ? w.frame_base[lane] : 0x7fffffff;
    for (int i = blockIdx.x * 4 + wid; i < count; i += gridDim.x * 4) {
        const int v = w.wide_list[i];
        const int f = __popcll(__ballot(lane >= 1 && lane < F && v >= fb));
        const int vl = v - __shfl(fb, f, 64);
        const long long dst = concat ? (long long)v : (long long)f * cap_voxels + vl;
        const int h = w.vox_slot[v];
        const int n = w.scount[(size_t)f * w.slots + h];
        gather_wide<C>(pcd, perm, n_points, cap, ncol, T, w, voxels, coords, counts, f, dst, h, n, concat, s_best, s_pts, wid, lane);
    }
}

}  // namespace

extern "C" int mvx_abi_version(void) { return 8; }

// fp16-piece operand scaling: see mvx_split_operand_amax in include/mvx_hip.h and split_common.h
static thread_local SplitAmax t_split_amax = {nullptr, nullptr, 0};
extern "C" int mvx_split_operand_amax(const float *amax_a, const float *amax_b) {
    t_split_amax = SplitAmax{amax_a, amax_b, 0};
    return MVX_OK;
}
void mvxi_drop_split_amax() { t_split_amax = SplitAmax{nullptr, nullptr, 0}; }
SplitAmax mvxi_take_split_amax() {
    const SplitAmax r = t_split_amax;
    t_split_amax = SplitAmax{nullptr, nullptr, 0};
    return r;
}

// Diagnostics: kernel launches issued through the library since it was loaded (the only process-wide state it keeps;
// hipMemsetAsync fills are not counted).  bench.py reports the difference over the timed steps.
static std::atomic<unsigned long long> g_launches{0};
void mvxi_count_launch() { g_launches.fetch_add(1, std::memory_order_relaxed); }
extern "C" uint64_t mvx_launch_count(void) { return g_launches.load(std::memory_order_relaxed); }

extern "C" size_t mvx_voxelize_workspace_bytes(int32_t n_frames, int32_t cap_points) {
    if (n_frames <= 0 || cap_points <= 0) return 0;
    size_t total = 0;
    carve(nullptr, n_frames, cap_points, &total);
    return total;
}

extern "C" int mvx_voxelize_frames(const float *pcd, const int32_t *perm, const int32_t *n_points,
                                   const int32_t *ext_idx, int32_t n_frames, int32_t cap_points, int32_t ncol,
                                   double lo_x, double lo_y, double lo_z,
                                   double size_x, double size_y, double size_z,
                                   int32_t T, int32_t out_channels, int32_t cap_voxels, int32_t concat,
                                   float *voxels, int64_t *coords, int32_t *counts, int32_t *n_voxels, int32_t *vox_off,
                                   int32_t *status, void *workspace, size_t workspace_bytes, void *stream) {
    MVX_CHECK_ARG(pcd && n_points && voxels && coords && counts && n_voxels && status && workspace);
    MVX_CHECK_ARG(n_frames > 0 && n_frames <= 255 && cap_points > 0 && cap_voxels > 0 && ncol >= 4);
    MVX_CHECK_ARG(T > 0 && T <= 64);
    MVX_CHECK_ARG(out_channels == 7 || out_channels == 9);
    MVX_CHECK_ARG(size_x > 0 && size_y > 0 && size_z > 0);
    if ((long long)cap_points > (1ll << 28) || (long long)cap_points * n_frames >= (1ll << 31)) return MVX_ESIZE;
    size_t need = 0;
    VoxWs w = carve(workspace, n_frames, cap_points, &need);
    MVX_CHECK_ARG(workspace_bytes >= need);
    hipStream_t st = (hipStream_t)stream;

    hipError_t e = hipMemsetAsync(w.keys, 0xFF, w.ff_bytes, st);             // empty keys, first = UINT_MAX
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(w.scount, 0, w.zero_bytes, st);                         // counts, scan ticket, look-back words
    if (e != hipSuccess) return (int)e;
    const unsigned gb = mvx_cdiv(cap_points, 256);
    hipLaunchKernelGGL(vox_insert, dim3(gb, n_frames), dim3(256), 0, st, pcd, perm, n_points, ext_idx, cap_points, ncol,
                       lo_x, lo_y, lo_z, size_x, size_y, size_z, w, status);
    MVX_LAUNCH_CHECK();
    hipLaunchKernelGGL(vox_scan, dim3(w.nblk), dim3(256), 0, st, n_points, cap_points, n_frames, w);
    MVX_LAUNCH_CHECK();
    const long long vmax = concat ? (cap_voxels < (long long)cap_points * n_frames ? cap_voxels : (long long)cap_points * n_frames)
                                  : (long long)cap_points * n_frames;
    const long long rounds16 = mvx_cdiv(vmax, 16); // four waves of four voxels per block and round
    const dim3 gg((unsigned)(rounds16 < 2048 ? rounds16 : 2048));
    if (out_channels == 9)
        hipLaunchKernelGGL(vox_gather<9>, gg, dim3(256), 16 * T * 9 * sizeof(float), st, pcd, perm, n_points, cap_points, ncol, T, n_frames, cap_voxels,
                           concat, w, voxels, (long long *)coords, counts, n_voxels, vox_off, status);
    else
        hipLaunchKernelGGL(vox_gather<7>, gg, dim3(256), 16 * T * 7 * sizeof(float), st, pcd, perm, n_points, cap_points, ncol, T, n_frames, cap_voxels,
                           concat, w, voxels, (long long *)coords, counts, n_voxels, vox_off, status);
    MVX_LAUNCH_CHECK();
    const dim3 gw((unsigned)(rounds16 < 512 ? rounds16 : 512));
    if (out_channels == 9)
        hipLaunchKernelGGL(vox_gather_wide<9>, gw, dim3(256), 0, st, pcd, perm, n_points, cap_points, ncol, T, n_frames, cap_voxels,
                           concat, w, voxels, (long long *)coords, counts);
    else
        hipLaunchKernelGGL(vox_gather_wide<7>, gw, dim3(256), 0, st, pcd, perm, n_points, cap_points, ncol, T, n_frames, cap_voxels,
                           concat, w, voxels, (long long *)coords, counts);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" int mvx_voxelize(const float *pcd, const int32_t *perm, const int32_t *n_points,
                            const int32_t *ext_idx, int32_t n_frames, int32_t cap_points, int32_t ncol,
                            double lo_x, double lo_y, double lo_z,
                            double size_x, double size_y, double size_z,
                            int32_t T, int32_t out_channels, int32_t cap_voxels,
                            float *voxels, int64_t *coords, int32_t *counts, int32_t *n_voxels,
                            int32_t *status, void *workspace, size_t workspace_bytes, void *stream) {
    return mvx_voxelize_frames(pcd, perm, n_points, ext_idx, n_frames, cap_points, ncol, lo_x, lo_y, lo_z, size_x, size_y,
                               size_z, T, out_channels, cap_voxels, 0, voxels, coords, counts, n_voxels, nullptr, status,
                               workspace, workspace_bytes, stream);
}
