/* Plain-C restatement of the reference's BEV rotated-box IoU and anchor classification --
 * TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).
 *
 * Follows cpp/voxelutil.cpp:
 *   :15-20   tolerance eps = 1e-6f, sig(d) = (d > eps) - (d < -eps)
 *   :28-38   cross product, shoelace area (f32 accumulation, one division by 2.0 in double)
 *   :39-48   intersection of line (a,b) with segment (c,d)
 *   :50-63   Sutherland-Hodgman cut of a polygon by the half plane left of (a,b), then removal of
 *            consecutive duplicates and of trailing copies of the first vertex
 *   :65-79   signed intersection area of the origin triangles (o,a,b) and (o,c,d)
 *   :81-93   polygon-polygon intersection = sum over edge pairs of the signed triangle terms
 *   :96-136  bboxOverlap / bboxIntersection.  The reference fills r2[j] (box index) instead of r2[k]
 *            (corner index) at :107-109,128-130 -- out-of-bounds for more than 5 boxes, garbage
 *            below that.  There is no defined behaviour to reproduce; this restatement (and the HIP
 *            kernel it checks) computes what the callers expect: IoU / intersection of box i with box j.
 *   :138-316 classifyAnchors: from the ground truth's centre cell walk rows up (h = 0,1,..) then down
 *            (h = -1,-2,..) until the centre-column IoU drops below 0.1; in every accepted row walk
 *            right (v = 1,2,..) then left (v = -1,-2,..) until IoU < 0.1.  IoU >= posThr -> positive
 *            (also listed among the non-negatives), IoU >= negThr -> non-negative only.
 *
 * One call-history dependence of the reference is kept deliberately: the cut routine's scratch polygon is
 * a function-local static (:51), and when the crossing test fails (|s2 - s1| <= eps although the end points
 * classify differently, :44) the scratch slot is consumed without being written, i.e. it keeps the value of
 * an earlier call.  This restatement keeps one static scratch as well, so sequential use reproduces the
 * reference exactly; `oracle_anchor_stale_reads` counts how often it happened (fixtures assert 0, because
 * a parallel implementation cannot reproduce call history).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

typedef struct { float x, y; } pt_t;

static const float TOL = 1e-6f;
static pt_t g_scratch[20];
static int64_t g_stale_reads = 0;

int64_t oracle_anchor_stale_reads(void) { return g_stale_reads; }

static int sgn(float d) { return (d > TOL) - (d < -TOL); }

static float cross3(pt_t o, pt_t a, pt_t b) {
    return (a.x - o.x) * (b.y - o.y) - (b.x - o.x) * (a.y - o.y);
}

/* ps must have room for n + 1 points: ps[n] is overwritten with ps[0] (as in the reference) */
static float shoelace(pt_t *ps, int n) {
    float acc = 0;
    ps[n] = ps[0];
    for (int i = 0; i < n; i++) acc += ps[i].x * ps[i + 1].y - ps[i].y * ps[i + 1].x;
    return (float)(acc / 2.0);
}

static int same_pt(pt_t p, pt_t q) { return sgn(p.x - q.x) == 0 && sgn(p.y - q.y) == 0; }

/* returns 1 and writes *out when the crossing exists; 0 / 2 leave *out untouched */
static int crossing(pt_t a, pt_t b, pt_t c, pt_t d, pt_t *out) {
    float s1 = cross3(a, b, c), s2 = cross3(a, b, d);
    if (sgn(s1) == 0 && sgn(s2) == 0) return 2;
    if (sgn(s2 - s1) == 0) return 0;
    out->x = (c.x * s2 - d.x * s1) / (s2 - s1);
    out->y = (c.y * s2 - d.y * s1) / (s2 - s1);
    return 1;
}

static void cut(pt_t *p, int *n_io, pt_t a, pt_t b) {
    int n = *n_io, m = 0;
    p[n] = p[0];
    for (int i = 0; i < n; i++) {
        int si = sgn(cross3(a, b, p[i])), sj = sgn(cross3(a, b, p[i + 1]));
        if (si > 0) g_scratch[m++] = p[i];
        if (si != sj) {
            if (crossing(a, b, p[i], p[i + 1], &g_scratch[m]) != 1) g_stale_reads++;
            m++;
        }
    }
    n = 0;
    for (int i = 0; i < m; i++)
        if (i == 0 || !same_pt(g_scratch[i], g_scratch[i - 1])) p[n++] = g_scratch[i];
    while (n > 1 && same_pt(p[n - 1], p[0])) n--;
    *n_io = n;
}

static float tri_pair(pt_t a, pt_t b, pt_t c, pt_t d) {
    pt_t o = {0.f, 0.f};
    int s1 = sgn(cross3(o, a, b)), s2 = sgn(cross3(o, c, d));
    if (s1 == 0 || s2 == 0) return 0.0f;
    if (s1 == -1) { pt_t t = a; a = b; b = t; }
    if (s2 == -1) { pt_t t = c; c = d; d = t; }
    pt_t p[10];
    p[0] = o; p[1] = a; p[2] = b;
    int n = 3;
    cut(p, &n, o, c);
    cut(p, &n, c, d);
    cut(p, &n, d, o);
    float res = (float)fabs(shoelace(p, n));
    return (s1 * s2 == -1) ? -res : res;
}

static void reverse_pts(pt_t *p, int n) {
    for (int i = 0, j = n - 1; i < j; i++, j--) { pt_t t = p[i]; p[i] = p[j]; p[j] = t; }
}

/* both arrays need room for 5 points; either may be reversed in place (reference :82-83) */
static float quad_intersection(pt_t *q1, pt_t *q2) {
    if (shoelace(q1, 4) < 0) reverse_pts(q1, 4);
    if (shoelace(q2, 4) < 0) reverse_pts(q2, 4);
    q1[4] = q1[0];
    q2[4] = q2[0];
    float res = 0;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) res += tri_pair(q1[i], q1[i + 1], q2[j], q2[j + 1]);
    return res;
}

static void load_quad(pt_t *q, const float *src) {
    for (int k = 0; k < 4; k++) { q[k].x = src[2 * k]; q[k].y = src[2 * k + 1]; }
}

/* b1 f32 [n][4][2], b2 f32 [m][4][2] -> out f32 [n][m]; iou != 0: IoU, else intersection area */
void oracle_bbox_pairwise(const float *b1, int64_t n, const float *b2, int64_t m, int iou, float *out) {
    pt_t q1[5], q2[5];
    for (int64_t i = 0; i < n; i++) {
        load_quad(q1, b1 + i * 8);
        float a1 = shoelace(q1, 4);
        for (int64_t j = 0; j < m; j++) {
            load_quad(q2, b2 + j * 8);
            float a2 = shoelace(q2, 4);
            float inter = quad_intersection(q1, q2);
            out[i * m + j] = iou ? inter / (a1 + a2 - inter) : inter;
        }
    }
}

typedef struct {
    int64_t *pos[3], *neg[3], *gi;
    int64_t n_pos, n_neg;
} lists_t;

/* one visited cell: returns 0 when the walk in this direction stops (IoU < 0.1) */
static int visit(const float *anchors, int64_t W, int64_t A, int64_t x, int64_t y, int64_t z, pt_t *gt, float gt_area,
                 float anchor_area, float neg_thr, float pos_thr, int64_t g, lists_t *o) {
    pt_t q[5];
    load_quad(q, anchors + ((x * W + y) * A + z) * 8);
    float inter = quad_intersection(gt, q);
    float iou = inter / (gt_area + anchor_area - inter);
    if (iou < 0.1) return 0;            /* double comparison against 0.1, as in the reference */
    const int is_pos = iou >= pos_thr;
    if (is_pos) {
        o->pos[0][o->n_pos] = x; o->pos[1][o->n_pos] = y; o->pos[2][o->n_pos] = z;
        o->gi[o->n_pos++] = g;
    }
    if (is_pos || iou >= neg_thr) {     /* every positive is listed among the non-negatives too */
        o->neg[0][o->n_neg] = x; o->neg[1][o->n_neg] = y; o->neg[2][o->n_neg] = z;
        o->n_neg++;
    }
    return 1;
}

static void walk_row(const float *anchors, int64_t W, int64_t A, int64_t x, int64_t ny, int64_t z, pt_t *gt, float gt_area,
                     float anchor_area, float neg_thr, float pos_thr, int64_t g, lists_t *o) {
    for (int64_t v = 1; ny + v < W; v++)
        if (!visit(anchors, W, A, x, ny + v, z, gt, gt_area, anchor_area, neg_thr, pos_thr, g, o)) break;
    for (int64_t v = -1; ny + v >= 0; v--)
        if (!visit(anchors, W, A, x, ny + v, z, gt, gt_area, anchor_area, neg_thr, pos_thr, g, o)) break;
}

/* gts f32 [G][4][2], anchors f32 [L][W][A][4][2], nls / nws i64 [G] (centre cells, inside the grid).
 * Output arrays sized by the caller (L*W*A*G entries always suffice).  counts[0] = positives, [1] = non-negatives. */
void oracle_classify_anchors(const float *gts, int64_t G, const float *anchors, int64_t L, int64_t W, int64_t A,
                             const int64_t *nls, const int64_t *nws, float neg_thr, float pos_thr,
                             int64_t *px, int64_t *py, int64_t *pz, int64_t *nx, int64_t *ny_, int64_t *nz, int64_t *gi,
                             int64_t *counts) {
    lists_t o = {{px, py, pz}, {nx, ny_, nz}, gi, 0, 0};
    pt_t q[5], gt[5];
    load_quad(q, anchors);
    float anchor_area = shoelace(q, 4);
    for (int64_t g = 0; g < G; g++) {
        int64_t nl = nls[g], nw = nws[g];
        load_quad(gt, gts + g * 8);
        float gt_area = shoelace(gt, 4);           /* signed, taken BEFORE any in-place reversal */
        for (int64_t z = 0; z < A; z++) {
            for (int64_t h = 0; nl + h < L; h++) {
                if (!visit(anchors, W, A, nl + h, nw, z, gt, gt_area, anchor_area, neg_thr, pos_thr, g, &o)) break;
                walk_row(anchors, W, A, nl + h, nw, z, gt, gt_area, anchor_area, neg_thr, pos_thr, g, &o);
            }
            for (int64_t h = -1; nl + h >= 0; h--) {
                if (!visit(anchors, W, A, nl + h, nw, z, gt, gt_area, anchor_area, neg_thr, pos_thr, g, &o)) break;
                walk_row(anchors, W, A, nl + h, nw, z, gt, gt_area, anchor_area, neg_thr, pos_thr, g, &o);
            }
        }
    }
    counts[0] = o.n_pos;
    counts[1] = o.n_neg;
}
