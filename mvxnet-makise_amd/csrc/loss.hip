// VoxelLoss forward + backward in two launches.
//
// Replaces modules/voxelnet/Loss.py:15-45 (and the autograd graph behind train.py:140,161):
//   posLoss = -sum_{pi} log(score + eps) / (n_pos + eps)
//   negLoss = (sum_all -log(1 - score + eps)  -  sum_{ni} -log(1 - score + eps)) / (N - n_neg + eps)
//   clsLoss = a posLoss + b negLoss
//   regLoss = mean over (n_pos x 7) of SmoothL1(reg[pi] - target),  target from the matched ground truth / anchor
// `ni` is the reference's list of NOT-negative anchors (every positive is in it too); an entry listed twice is
// subtracted twice, exactly like the indexed sum of the reference.  Sums run in f64; the two denominators are rounded
// to f32 the way `tensor / python_float` rounds them.
//
// The score / regression maps are read and their gradients written through explicit strides, so the RPN's
// (1,2,L,W) / (1,14,L,W) outputs are used as they are (train.py:132-133 only permutes views).
#include "common.h"

namespace {

struct Strides { long long l, w, c; };

// pass 1: every score.  d_score = g_cls * b / (1 - s + eps) / den_neg ; partial sums of -log(1 - s + eps)
__global__ void loss_dense(const float *__restrict__ score, Strides ss, int L, int W, int A, float eps, float scale_neg,
                           float *__restrict__ dscore, Strides ds, double *__restrict__ acc) {
    const long long N = (long long)L * W * A;
    double part = 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x) {
        // iterate in MEMORY order of the common (1,A,L,W) layout: a = slowest
        const int a = (int)(i / ((long long)L * W));
        const long long r = i % ((long long)L * W);
        const int x = (int)(r / W), y = (int)(r % W);
        const float s = score[x * ss.l + y * ss.w + a * ss.c];
        const float om = 1.f - s + eps;
        part += (double)(-logf(om));
        if (dscore) dscore[x * ds.l + y * ds.w + a * ds.c] = scale_neg / om;
    }
    part = wave_sum_f64(part);
    __shared__ double s_part[16];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) s_part[wid] = part;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) t += s_part[k];
        atomicAdd(acc, t);
    }
}

// pass 2 (one workgroup): the listed anchors, then the two scalars.
// acc f64 [4]: [0] = sum_all (from pass 1), [1] = sum over ni, [2] = sum over pi of -log(s + eps), [3] = SmoothL1 sum
__global__ void loss_lists(const float *__restrict__ score, Strides ss, const float *__restrict__ reg, Strides rs,
                           const long long *__restrict__ pos_idx, long long pos_ld, const long long *__restrict__ neg_idx,
                           long long neg_ld, const long long *__restrict__ gi, const int *__restrict__ counts_dev,
                           int n_pos_host, int n_neg_host, const float *__restrict__ gts, int gt_ld,
                           const float *__restrict__ anchors, int L, int W, int A, float a_w, float b_w, float eps,
                           float *__restrict__ dscore, Strides ds, float *__restrict__ dreg, Strides dr,
                           double *__restrict__ acc, float *__restrict__ losses) {
    const int n_pos = counts_dev ? counts_dev[0] : n_pos_host, n_neg = counts_dev ? counts_dev[1] : n_neg_host;
    const long long N = (long long)L * W * A;
    const float den_pos = (float)((double)n_pos + (double)eps);
    const float den_neg = (float)((double)(N - n_neg) + (double)eps);
    double s_neg = 0.0, s_pos = 0.0, s_reg = 0.0;
    for (int k = threadIdx.x; k < n_neg; k += blockDim.x) {
        const long long x = neg_idx[k], y = neg_idx[neg_ld + k], z = neg_idx[2 * neg_ld + k];
        const float om = 1.f - score[x * ss.l + y * ss.w + z * ss.c] + eps;
        s_neg += (double)(-logf(om));
        if (dscore) atomicAdd(dscore + x * ds.l + y * ds.w + z * ds.c, -(b_w / den_neg) / om);
    }
    const float reg_scale = n_pos > 0 ? 1.f / (float)(n_pos * 7) : 0.f;
    for (int k = threadIdx.x; k < n_pos; k += blockDim.x) {
        const long long x = pos_idx[k], y = pos_idx[pos_ld + k], z = pos_idx[2 * pos_ld + k];
        const float sp = score[x * ss.l + y * ss.w + z * ss.c] + eps;
        s_pos += (double)(-logf(sp));
        if (dscore) atomicAdd(dscore + x * ds.l + y * ds.w + z * ds.c, -(a_w / den_pos) / sp);
        if (!reg) continue;
        const float *g = gts + (size_t)gi[k] * gt_ld;
        const float *an = anchors + ((size_t)(x * W + y) * A + z) * 7;
        const float d = sqrtf(an[3] * an[3] + an[4] * an[4]);
        float t[7];
        t[0] = (g[0] - an[0]) / d;
        t[1] = (g[1] - an[1]) / d;
        t[2] = (g[2] - an[2]) / an[5];
        t[3] = logf(g[3] / an[3]);
        t[4] = logf(g[4] / an[4]);
        t[5] = logf(g[5] / an[5]);
        t[6] = g[6] - an[6];
#pragma unroll
        for (int c = 0; c < 7; ++c) {
            const long long off_c = z * 7 + c;
            const float diff = reg[x * rs.l + y * rs.w + off_c * rs.c] - t[c];
            const float ad = fabsf(diff);
            s_reg += (double)(ad < 1.f ? 0.5f * diff * diff : ad - 0.5f);
            if (dreg) atomicAdd(dreg + x * dr.l + y * dr.w + off_c * dr.c, reg_scale * (ad < 1.f ? diff : (diff > 0.f ? 1.f : -1.f)));
        }
    }
    __shared__ double s_red[3][16];
    s_neg = wave_sum_f64(s_neg);
    s_pos = wave_sum_f64(s_pos);
    s_reg = wave_sum_f64(s_reg);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) { s_red[0][wid] = s_neg; s_red[1][wid] = s_pos; s_red[2][wid] = s_reg; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double tn = 0.0, tp = 0.0, tr = 0.0;
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) { tn += s_red[0][k]; tp += s_red[1][k]; tr += s_red[2][k]; }
        acc[1] = tn; acc[2] = tp; acc[3] = tr;
        const float pos_loss = (float)tp / den_pos;
        const float neg_loss = (float)(acc[0] - tn) / den_neg;
        losses[0] = a_w * pos_loss + b_w * neg_loss;
        losses[1] = n_pos > 0 ? (float)(tr / (double)(n_pos * 7)) : 0.f;
    }
}

}  // namespace

extern "C" int mvx_voxel_loss(const float *score, int64_t score_sl, int64_t score_sw, int64_t score_sa, const float *reg,
                              int64_t reg_sl, int64_t reg_sw, int64_t reg_sc, const int64_t *pos_idx, int64_t pos_ld,
                              const int64_t *neg_idx, int64_t neg_ld, const int64_t *gi, const int32_t *counts_dev,
                              int32_t n_pos, int32_t n_neg, const float *gts, int32_t gt_ld, const float *anchors, int32_t l,
                              int32_t w, int32_t anchors_per_loc, float a, float b, float eps, float *dscore,
                              int64_t dscore_sl, int64_t dscore_sw, int64_t dscore_sa, float *dreg, int64_t dreg_sl,
                              int64_t dreg_sw, int64_t dreg_sc, float *losses, double *scratch, void *stream) {
    MVX_CHECK_ARG(score && losses && scratch && l > 0 && w > 0 && anchors_per_loc > 0);
    MVX_CHECK_ARG(n_pos >= 0 && n_neg >= 0);
    MVX_CHECK_ARG(counts_dev || ((n_pos == 0 || pos_idx) && (n_neg == 0 || neg_idx)));
    MVX_CHECK_ARG(!(reg && (n_pos > 0 || counts_dev)) || (gi && gts && anchors && gt_ld >= 7));
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(scratch, 0, 4 * sizeof(double), st);
    if (e != hipSuccess) return (int)e;
    const long long N = (long long)l * w * anchors_per_loc;
    // the dense pass needs the number of listed non-negatives for its denominator: with device-side counts the gradient
    // scale is applied in the list pass instead (dscore_dense = b / om, rescaled there) -- not supported: the host knows
    // the counts whenever it asks for gradients
    MVX_CHECK_ARG(!(counts_dev && dscore));
    const float den_neg = (float)((double)(N - n_neg) + (double)eps);
    const Strides ss = {score_sl, score_sw, score_sa}, rs = {reg_sl, reg_sw, reg_sc};
    const Strides ds = {dscore_sl, dscore_sw, dscore_sa}, dr = {dreg_sl, dreg_sw, dreg_sc};
    unsigned blocks = mvx_cdiv(N, 256 * 4);
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(loss_dense, dim3(blocks), dim3(256), 0, st, score, ss, l, w, anchors_per_loc, eps, b / den_neg, dscore, ds,
                       scratch);
    MVX_LAUNCH_CHECK();
    hipLaunchKernelGGL(loss_lists, dim3(1), dim3(1024), 0, st, score, ss, reg, rs, (const long long *)pos_idx, (long long)pos_ld,
                       (const long long *)neg_idx, (long long)neg_ld, (const long long *)gi, counts_dev, n_pos, n_neg, gts, gt_ld,
                       anchors, l, w, anchors_per_loc, a, b, eps, dscore, ds, dreg, dr, scratch, losses);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}
