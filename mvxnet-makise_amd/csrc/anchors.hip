// BEV rotated-box IoU and anchor classification on the GPU.
//
// Replaces the CPU label preparation of the reference extension: cpp/voxelutil.cpp:96-136 (bboxOverlap,
// bboxIntersection) and :138-316 (classifyAnchors), called from modules/Calc.py:88-96 and
// modules/augment/Augment.py:54.  The arithmetic is the reference's, operation for operation in f32
// (origin-fan triangulation, half-plane cuts with the 1e-6 tolerance, shoelace areas halved in f64), so the
// integer results (which anchors are positive / non-negative, in which order) are bit-identical; the library
// is compiled with -ffp-contract=off.
//
// Parallel form of classifyAnchors: the reference walks outwards from the ground truth's centre cell until the IoU
// drops below 0.1.  Here one workgroup per (ground truth, anchor orientation) evaluates the IoU of every cell of a
// (2R+1)^2 window around the centre in parallel into LDS, then one lane replays the reference's walk over the
// window (rows up, rows down; inside a row right, then left) -- same visiting order, same break conditions -- and
// a second single-workgroup launch concatenates the per-pair lists in (ground truth, orientation) order.
#include "common.h"

namespace {

struct P2 { float x, y; };

// The polygons of the clipping live in LDS, one slot column per thread (element i of thread t at base[i * stride + t]:
// consecutive lanes hit consecutive 8-byte words, no bank conflicts).  Thread-private arrays indexed by run-time counters
// (q[m++]) are placed in scratch memory by the compiler: 336 B per thread and ~2,000 dependent scratch accesses per IoU
// made one classifyAnchors call of 8 boxes take 0.5 ms.
struct LP {
    P2 *b;
    int stride;
    __device__ __forceinline__ P2 &operator[](int i) const { return b[i * stride]; }
};
constexpr int POLY_SLOTS = 40;          // per thread: p[10] | q[20] | quad 1 [5] | quad 2 [5]

constexpr float TOL = 1e-6f;

__device__ __forceinline__ int sgn(float d) { return (d > TOL) - (d < -TOL); }

__device__ __forceinline__ float cross3(P2 o, P2 a, P2 b) { return (a.x - o.x) * (b.y - o.y) - (b.x - o.x) * (a.y - o.y); }

__device__ __forceinline__ bool same_pt(P2 p, P2 q) { return sgn(p.x - q.x) == 0 && sgn(p.y - q.y) == 0; }

// shoelace area of ps[0..n) (ps[n] is set to ps[0]); f32 accumulation, the halving in f64 (voxelutil.cpp:31-38)
__device__ float shoelace(LP ps, int n) {
    float acc = 0.f;
    ps[n] = ps[0];
    for (int i = 0; i < n; ++i) {
        const P2 u = ps[i], v = ps[i + 1];
        acc += u.x * v.y - u.y * v.x;
    }
    return (float)((double)acc / 2.0);
}

// polygon p[0..n) cut by the half plane left of (a, b) (voxelutil.cpp:50-63); q: 20 slots of scratch
__device__ void cut(LP p, int &n, P2 a, P2 b, LP q) {
    int m = 0;
    p[n] = p[0];
    for (int i = 0; i < n; ++i) {
        const P2 pi = p[i], pj = p[i + 1];
        const float s1 = cross3(a, b, pi), s2 = cross3(a, b, pj);
        const int g1 = sgn(s1), g2 = sgn(s2);
        if (g1 > 0) q[m++] = pi;
        if (g1 != g2) {
            // The reference consumes a slot even when |s2 - s1| <= 1e-6 makes it skip the crossing (voxelutil.cpp:44),
            // leaving whatever an EARLIER call stored there; call history does not exist here, the slot takes p[i].
            P2 c = pi;
            if (sgn(s2 - s1) != 0) {
                c.x = (pi.x * s2 - pj.x * s1) / (s2 - s1);
                c.y = (pi.y * s2 - pj.y * s1) / (s2 - s1);
            }
            q[m++] = c;
        }
    }
    n = 0;
    for (int i = 0; i < m; ++i)
        if (i == 0 || !same_pt(q[i], q[i - 1])) p[n++] = q[i];
    while (n > 1 && same_pt(p[n - 1], p[0])) --n;
}

// signed intersection area of the origin triangles (o,a,b) and (o,c,d) (voxelutil.cpp:65-79); p: 10 slots, q: 20 slots
__device__ float tri_pair(P2 a, P2 b, P2 c, P2 d, LP p, LP q) {
    const P2 o = {0.f, 0.f};
    const int s1 = sgn(cross3(o, a, b)), s2 = sgn(cross3(o, c, d));
    if (s1 == 0 || s2 == 0) return 0.f;
    if (s1 == -1) { const P2 t = a; a = b; b = t; }
    if (s2 == -1) { const P2 t = c; c = d; d = t; }
    p[0] = o; p[1] = a; p[2] = b;
    int n = 3;
    cut(p, n, o, c, q);
    cut(p, n, c, d, q);
    cut(p, n, d, o, q);
    const float res = (float)fabs((double)shoelace(p, n));
    return (s1 * s2 == -1) ? -res : res;
}

__device__ void orient_ccw(LP q) {      // voxelutil.cpp:82-83
    if (shoelace(q, 4) < 0.f) {
        P2 t = q[0]; q[0] = q[3]; q[3] = t;
        t = q[1]; q[1] = q[2]; q[2] = t;
    }
    q[4] = q[0];
}

// q1, q2: 5 slots each, both already oriented (orient_ccw)
__device__ float quad_intersection(LP q1, LP q2, LP p, LP q) {
    float res = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) res += tri_pair(q1[i], q1[i + 1], q2[j], q2[j + 1], p, q);
    return res;
}

__device__ __forceinline__ void load_quad(LP q, const float *src) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { P2 v; v.x = src[2 * k]; v.y = src[2 * k + 1]; q[k] = v; }
}

// the four polygon areas of this thread inside a [POLY_SLOTS][threads] LDS block
struct Polys { LP p, q, q1, q2; };
__device__ __forceinline__ Polys polys_of(P2 *block, int threads, int t) {
    Polys r;
    r.p = LP{block + t, threads};
    r.q = LP{block + 10 * threads + t, threads};
    r.q1 = LP{block + 30 * threads + t, threads};
    r.q2 = LP{block + 35 * threads + t, threads};
    return r;
}

constexpr int PAIR_THREADS = 64;
__global__ __launch_bounds__(PAIR_THREADS) void bbox_pairwise(const float *__restrict__ b1, int n, const float *__restrict__ b2, int m,
                                                              int iou, float *__restrict__ out) {
    __shared__ P2 s_poly[POLY_SLOTS * PAIR_THREADS];
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)n * m) return;
    const int i = (int)(t / m), j = (int)(t % m);
    const Polys w = polys_of(s_poly, PAIR_THREADS, threadIdx.x);
    load_quad(w.q1, b1 + (size_t)i * 8);
    load_quad(w.q2, b2 + (size_t)j * 8);
    const float a1 = shoelace(w.q1, 4), a2 = shoelace(w.q2, 4);     // signed, before the re-orientation (as the reference)
    orient_ccw(w.q1);
    orient_ccw(w.q2);
    const float inter = quad_intersection(w.q1, w.q2, w.p, w.q);
    out[t] = iou ? inter / (a1 + a2 - inter) : inter;
}

// One workgroup per (ground truth g, orientation z).  LDS: (2R+1)^2 IoUs.
// lists: per pair `cap_pair` i32 entries each for positives and non-negatives (flat cell index x*W + y).
__global__ __launch_bounds__(256) void anchor_window_walk(const float *__restrict__ gts, const float *__restrict__ anchors, int L, int W, int A,
                                   const long long *__restrict__ nls, const long long *__restrict__ nws, float neg_thr,
                                   float pos_thr, int R, int *__restrict__ pair_counts, int *__restrict__ pos_list,
                                   int *__restrict__ neg_list, int cap_pair, int *__restrict__ status) {
    extern __shared__ float s_iou[];
    __shared__ P2 s_poly[POLY_SLOTS * 256];
    const int pair = blockIdx.x, g = pair / A, z = pair % A;
    const int Wn = 2 * R + 1;
    const long long nl = nls[g], nw = nws[g];
    if (nl < 0 || nl >= L || nw < 0 || nw >= W) {     // the reference reads out of bounds here; skipped and reported
        if (threadIdx.x == 0) {
            pair_counts[2 * pair] = 0;
            pair_counts[2 * pair + 1] = 0;
            atomicOr(status, 4);
        }
        return;
    }
    const Polys wk = polys_of(s_poly, 256, threadIdx.x);
    const LP gt = wk.q1, q = wk.q2;
    load_quad(gt, gts + (size_t)g * 8);
    const float gt_area = shoelace(gt, 4);
    orient_ccw(gt);
    load_quad(q, anchors);
    const float anchor_area = shoelace(q, 4);
    // Bounding circles: boxes whose centres are further apart than the two half diagonals (+ 1 %) cannot touch.  Their
    // true IoU is 0 and the reference's origin-fan sum gives rounding noise of ~1e-6 there; either ends the walk
    // (iou < 0.1) the same way and neither value is ever output, so those cells skip the clipping.
    P2 gc = {0.f, 0.f};
    float gr = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const P2 v = gt[k]; gc.x += 0.25f * v.x; gc.y += 0.25f * v.y; }
#pragma unroll
    for (int k = 0; k < 4; ++k) { const P2 v = gt[k]; gr = fmaxf(gr, sqrtf((v.x - gc.x) * (v.x - gc.x) + (v.y - gc.y) * (v.y - gc.y))); }
    for (int c = threadIdx.x; c < Wn * Wn; c += blockDim.x) {
        const long long x = nl + c / Wn - R, y = nw + c % Wn - R;
        float iou = -1.f;                             // outside the grid: never visited (loop bounds of the reference)
        if (x >= 0 && x < L && y >= 0 && y < W) {
            load_quad(q, anchors + ((size_t)(x * W + y) * A + z) * 8);
            P2 ac = {0.f, 0.f};
            float ar = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) { const P2 v = q[k]; ac.x += 0.25f * v.x; ac.y += 0.25f * v.y; }
#pragma unroll
            for (int k = 0; k < 4; ++k) { const P2 v = q[k]; ar = fmaxf(ar, sqrtf((v.x - ac.x) * (v.x - ac.x) + (v.y - ac.y) * (v.y - ac.y))); }
            const float dist = sqrtf((ac.x - gc.x) * (ac.x - gc.x) + (ac.y - gc.y) * (ac.y - gc.y));
            if (dist > 1.01f * (gr + ar) + 1e-3f) {
                iou = 0.f;
            } else {
                orient_ccw(q);
                const float inter = quad_intersection(gt, q, wk.p, wk.q);
                iou = inter / (gt_area + anchor_area - inter);
            }
        }
        s_iou[c] = iou;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    int np = 0, nn = 0, overflow = 0;
    int *pl = pos_list + (size_t)pair * cap_pair, *nlst = neg_list + (size_t)pair * cap_pair;
    // visit(dh, dv): false = the walk in this direction ends
    auto visit = [&](int dh, int dv) -> bool {
        const long long x = nl + dh, y = nw + dv;
        if (x < 0 || x >= L || y < 0 || y >= W) return false;
        if (dh < -R || dh > R || dv < -R || dv > R) { overflow = 1; return false; }
        const float iou = s_iou[(dh + R) * Wn + (dv + R)];
        if ((double)iou < 0.1) return false;
        const bool is_pos = iou >= pos_thr;
        const int cell = (int)(x * W + y);
        if (is_pos) pl[np++] = cell;
        if (is_pos || iou >= neg_thr) nlst[nn++] = cell;
        return true;
    };
    for (int dir = 0; dir < 2; ++dir) {               // rows upwards from the centre row, then downwards from the one below
        for (int h = dir ? -1 : 0;; h += dir ? -1 : 1) {
            if (!visit(h, 0)) break;
            for (int v = 1; visit(h, v); ++v) {}
            for (int v = -1; visit(h, v); --v) {}
        }
    }
    pair_counts[2 * pair] = np;
    pair_counts[2 * pair + 1] = nn;
    if (overflow) atomicOr(status, 1);
}

// One workgroup per FRAME: concatenates the per-pair lists of the frame's ground truths in pair order into the i64 index
// triples of the reference.  ``goff`` = ground-truth offsets of the frames (a single frame: {0, n_gt}); frame f writes
// pos_idx[f][3][cap], neg_idx[f][3][cap], gi[f][cap] (ground-truth ids LOCAL to the frame) and counts[f][2].
struct GtOffsets { int off[MVX_MAX_FRAMES + 1]; };

__global__ void anchor_concat(const int *__restrict__ pair_counts, const int *__restrict__ pos_list,
                              const int *__restrict__ neg_list, GtOffsets goff, int A, int W, int cap_pair,
                              long long *__restrict__ pos_idx, long long *__restrict__ neg_idx, long long *__restrict__ gi,
                              long long cap, int *__restrict__ counts, int *__restrict__ status) {
    __shared__ int s_scan[17];
    __shared__ int s_base[2];
    const int f = blockIdx.x;
    const int g_lo = goff.off[f], pair_lo = g_lo * A, pair_hi = goff.off[f + 1] * A;
    pos_idx += (size_t)f * 3 * cap;
    neg_idx += (size_t)f * 3 * cap;
    gi += (size_t)f * cap;
    counts += 2 * f;
    if (threadIdx.x == 0) { s_base[0] = 0; s_base[1] = 0; }
    __syncthreads();
    for (int p0 = pair_lo; p0 < pair_hi; p0 += blockDim.x) {
        const int p = p0 + threadIdx.x;
        const int cp = p < pair_hi ? pair_counts[2 * p] : 0, cn = p < pair_hi ? pair_counts[2 * p + 1] : 0;
        int tot_p, tot_n;
        const int op = block_excl_scan_i32(cp, s_scan, &tot_p) + s_base[0];
        const int on = block_excl_scan_i32(cn, s_scan, &tot_n) + s_base[1];
        if (p < pair_hi) {
            const long long g = p / A - g_lo, z = p % A;
            for (int k = 0; k < cp; ++k) {
                const long long o = op + k;
                if (o >= cap) { atomicOr(status, 2); break; }
                const int cell = pos_list[(size_t)p * cap_pair + k];
                pos_idx[o] = cell / W;
                pos_idx[cap + o] = cell % W;
                pos_idx[2 * cap + o] = z;
                gi[o] = g;
            }
            for (int k = 0; k < cn; ++k) {
                const long long o = on + k;
                if (o >= cap) { atomicOr(status, 2); break; }
                const int cell = neg_list[(size_t)p * cap_pair + k];
                neg_idx[o] = cell / W;
                neg_idx[cap + o] = cell % W;
                neg_idx[2 * cap + o] = z;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) { s_base[0] += tot_p; s_base[1] += tot_n; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        counts[0] = s_base[0] < cap ? s_base[0] : (int)cap;
        counts[1] = s_base[1] < cap ? s_base[1] : (int)cap;
    }
}

}  // namespace

extern "C" int mvx_bbox_pairwise(const float *boxes1, int32_t n1, const float *boxes2, int32_t n2, int32_t want_iou,
                                 float *out, void *stream) {
    MVX_CHECK_ARG(n1 >= 0 && n2 >= 0);
    if (n1 == 0 || n2 == 0) return MVX_OK;
    MVX_CHECK_ARG(boxes1 && boxes2 && out);
    const long long tot = (long long)n1 * n2;
    hipLaunchKernelGGL(bbox_pairwise, dim3(mvx_cdiv(tot, PAIR_THREADS)), dim3(PAIR_THREADS), 0, (hipStream_t)stream, boxes1, n1, boxes2, n2,
                       want_iou, out);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" size_t mvx_classify_anchors_workspace_bytes(int32_t n_gt, int32_t anchors_per_loc, int32_t window_radius) {
    const size_t pairs = (size_t)(n_gt > 0 ? n_gt : 0) * (size_t)(anchors_per_loc > 0 ? anchors_per_loc : 0);
    const size_t wn = 2 * (size_t)window_radius + 1;
    return pairs * (2 + 2 * wn * wn) * sizeof(int32_t) + 256;
}

extern "C" int mvx_classify_anchors_frames(const float *gts, const int32_t *gt_off_host, int32_t n_frames, const float *anchors,
                                           int32_t l, int32_t w, int32_t anchors_per_loc, const int64_t *nls, const int64_t *nws,
                                           float neg_thr, float pos_thr, int32_t window_radius, int64_t *pos_idx,
                                           int64_t *neg_idx, int64_t *gi, int64_t cap, int32_t *counts, int32_t *status,
                                           void *workspace, size_t workspace_bytes, void *stream) {
    MVX_CHECK_ARG(gt_off_host && n_frames >= 1 && n_frames <= MVX_MAX_FRAMES);
    MVX_CHECK_ARG(l > 0 && w > 0 && anchors_per_loc > 0 && window_radius >= 1 && window_radius <= 55);
    MVX_CHECK_ARG(counts && status && cap >= 0);
    MVX_CHECK_ARG((long long)l * w < (1ll << 31));
    GtOffsets goff;
    for (int f = 0; f <= MVX_MAX_FRAMES; ++f) goff.off[f] = gt_off_host[f < n_frames ? f : n_frames];
    MVX_CHECK_ARG(goff.off[0] == 0);
    for (int f = 0; f < n_frames; ++f) MVX_CHECK_ARG(goff.off[f + 1] >= goff.off[f]);
    const int n_gt = goff.off[n_frames];
    hipStream_t st = (hipStream_t)stream;
    const int pairs = n_gt * anchors_per_loc;
    if (pairs == 0) {
        hipError_t e = hipMemsetAsync(counts, 0, 2 * sizeof(int32_t) * n_frames, st);
        return e == hipSuccess ? MVX_OK : (int)e;
    }
    MVX_CHECK_ARG(gts && anchors && nls && nws && pos_idx && neg_idx && gi && workspace);
    MVX_CHECK_ARG(workspace_bytes >= mvx_classify_anchors_workspace_bytes(n_gt, anchors_per_loc, window_radius));
    const int wn = 2 * window_radius + 1, cap_pair = wn * wn;
    int *pair_counts = (int *)workspace;
    int *pos_list = pair_counts + 2 * (size_t)pairs;
    int *neg_list = pos_list + (size_t)pairs * cap_pair;
    // ONE walk launch for the ground truths of all frames (a workgroup per (ground truth, orientation) pair) ...
    hipLaunchKernelGGL(anchor_window_walk, dim3(pairs), dim3(256), (size_t)cap_pair * sizeof(float), st, gts, anchors, l, w,
                       anchors_per_loc, (const long long *)nls, (const long long *)nws, neg_thr, pos_thr, window_radius,
                       pair_counts, pos_list, neg_list, cap_pair, status);
    MVX_LAUNCH_CHECK();
    // ... and one concatenation workgroup per frame
    hipLaunchKernelGGL(anchor_concat, dim3(n_frames), dim3(1024), 0, st, pair_counts, pos_list, neg_list, goff, anchors_per_loc, w,
                       cap_pair, (long long *)pos_idx, (long long *)neg_idx, (long long *)gi, (long long)cap, counts, status);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" int mvx_classify_anchors(const float *gts, int32_t n_gt, const float *anchors, int32_t l, int32_t w,
                                    int32_t anchors_per_loc, const int64_t *nls, const int64_t *nws, float neg_thr,
                                    float pos_thr, int32_t window_radius, int64_t *pos_idx, int64_t *neg_idx, int64_t *gi,
                                    int64_t cap, int32_t *counts, int32_t *status, void *workspace, size_t workspace_bytes,
                                    void *stream) {
    MVX_CHECK_ARG(n_gt >= 0);
    const int32_t off[2] = {0, n_gt};
    return mvx_classify_anchors_frames(gts, off, 1, anchors, l, w, anchors_per_loc, nls, nws, neg_thr, pos_thr, window_radius,
                                       pos_idx, neg_idx, gi, cap, counts, status, workspace, workspace_bytes, stream);
}
