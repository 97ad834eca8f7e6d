// BatchNorm with batch statistics (no affine, biased variance) on channels-last rows, forward
// and backward, fused with the ReLU that precedes it in every block of the reference
// (modules/layers/Blocks.py:14-16 and :28-29: Linear/Conv -> ReLU -> BN).
//
// All kernels see a row-major [rows][C] f32 matrix (C % 4 == 0).  Reductions keep f32
// partials per thread, f64 across the block and f64 atomics across blocks, so the batch
// statistics do not depend on the launch geometry beyond f64 rounding.
#include "common.h"
#include "split_common.h"

namespace {

constexpr int REP = 32;   // replicas of the cross-block accumulators (spreads same-address atomics)

// stats[0][C] = sum, stats[1][C] = sum of squares  ->  mi[0][C] = mean, mi[1][C] = 1/sqrt(var+eps)
__global__ void bn_finalize(const double *__restrict__ stats, double count, double eps, float *__restrict__ mi, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    stats += (size_t)blockIdx.y * MVX_REP * 2 * C;      // frame
    mi += (size_t)blockIdx.y * 2 * C;
    double s1 = 0.0, s2 = 0.0;
    for (int rp = 0; rp < MVX_REP; ++rp) {
        s1 += stats[((size_t)rp * 2) * C + c];
        s2 += stats[((size_t)rp * 2 + 1) * C + c];
    }
    const double mean = s1 / count;
    double var = s2 / count - mean * mean;
    if (var < 0.0) var = 0.0;
    mi[c] = (float)mean;
    mi[C + c] = (float)(1.0 / sqrt(var + eps));
}

__global__ void bn_apply(const float *__restrict__ y, const float *__restrict__ mi, float *__restrict__ out,
                         size_t n4, int C, FrameMap fm) {
    const int c4 = C >> 2;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < n4; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % c4) * 4;
        const float *fmi = mi + (size_t)fm_frame_of(fm, (long long)(e / c4)) * 2 * C;
        const float4 v = ((const float4 *)y)[e];
        const float4 m = *(const float4 *)(fmi + c), s = *(const float4 *)(fmi + C + c);
        float4 o;
        o.x = (v.x - m.x) * s.x; o.y = (v.y - m.y) * s.y; o.z = (v.z - m.z) * s.z; o.w = (v.w - m.w) * s.w;
        ((float4 *)out)[e] = o;
    }
}

// per-channel sum / sum of squares of a [rows][C] matrix (used when the producer did not
// already reduce them in its epilogue)
__global__ __launch_bounds__(256) void row_stats(const float *__restrict__ y, double *__restrict__ stats, size_t rows, int C) {
    __shared__ double red[2][256][4];
    const int c4 = C >> 2;
    const int rpi = max(1, 256 / c4);                 // rows per block iteration
    const int ct = threadIdx.x % c4, rt = threadIdx.x / c4;
    for (int cb = 0; cb < c4; cb += 256) {             // C > 1024 -> several column blocks
        const int col = cb + ct;
        float4 s1 = make_float4(0, 0, 0, 0), s2 = make_float4(0, 0, 0, 0);
        if (rt < rpi && col < c4) {
            for (size_t r = blockIdx.x * (size_t)rpi + rt; r < rows; r += (size_t)gridDim.x * rpi) {
                const float4 v = *(const float4 *)(y + r * C + col * 4);
                s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
                s2.x += v.x * v.x; s2.y += v.y * v.y; s2.z += v.z * v.z; s2.w += v.w * v.w;
            }
        }
        red[0][threadIdx.x][0] = s1.x; red[0][threadIdx.x][1] = s1.y; red[0][threadIdx.x][2] = s1.z; red[0][threadIdx.x][3] = s1.w;
        red[1][threadIdx.x][0] = s2.x; red[1][threadIdx.x][1] = s2.y; red[1][threadIdx.x][2] = s2.z; red[1][threadIdx.x][3] = s2.w;
        __syncthreads();
        if (rt == 0 && col < c4) {
            for (int k = 0; k < 2; ++k)
                for (int j = 0; j < 4; ++j) {
                    double t = 0.0;
                    for (int r = 0; r < rpi; ++r) t += red[k][r * c4 + ct][j];
                    atomicAdd(stats + ((size_t)(blockIdx.x % MVX_REP) * 2 + k) * C + col * 4 + j, t);
                }
        }
        __syncthreads();
    }
}

// backward, pass 1: sums[f][rep][0][c] = sum dyh, [1][c] = sum dyh * yhat   (yhat = (y - mean) * inv), per frame f.
// A block owns a CONTIGUOUS run of rows (rows_per_block) and makes one reduction pass per row segment it meets
// (almost always one: segments are whole frames).
// TRIP rows per thread per trip = 2 * TRIP independent 16-byte loads in flight (the pass is pure streaming: what limits it is
// the number of bytes in flight per CU).  The workgroup that finishes last folds the replicas of every frame into
// ab[f][0][c] = sum dyh / count, ab[f][1][c] = sum dyh * yhat / count, so that pass 2 starts from two floats per channel
// instead of 2 * REP device-scope f64 reads per thread (atomics-only hand-off as in bn_finalize_by_last_block, common.h).
template <int TRIP>
__global__ __launch_bounds__(256) void bn_bwd_reduce(const float *__restrict__ dyh, const float *__restrict__ y,
                                                     const float *__restrict__ mi, double *__restrict__ sums,
                                                     size_t rows, int C, size_t rows_per_block, FrameMap fm,
                                                     unsigned *__restrict__ done_counter, float *__restrict__ ab,
                                                     unsigned *__restrict__ amax_slot) {
    __shared__ double red[2][256][4];
    __shared__ int s_last;
    if (amax_slot && blockIdx.x == 0 && threadIdx.x == 0) *amax_slot = 0u;      // max |dz| of the apply pass starts from zero
    const int c4 = C >> 2;
    const int rpi = max(1, 256 / c4);
    const int ct = threadIdx.x % c4, rt = threadIdx.x / c4;
    const size_t blk_lo = blockIdx.x * rows_per_block;
    const size_t blk_hi = blk_lo + rows_per_block < rows ? blk_lo + rows_per_block : rows;
    if (blk_lo < blk_hi) {
    const int s_lo = fm.F == 1 ? 0 : fm_seg_of(fm, (long long)blk_lo), s_hi = fm.F == 1 ? 0 : fm_seg_of(fm, (long long)blk_hi - 1);
    for (int sg = s_lo; sg <= s_hi; ++sg) {
        const int f = fm.F == 1 ? 0 : (int)fm.seg_frame[sg];
        const size_t lo = fm.F == 1 ? blk_lo : max(blk_lo, (size_t)fm.bound[sg]);
        const size_t hi = fm.F == 1 ? blk_hi : min(blk_hi, (size_t)fm.bound[sg + 1]);
        const float *fmi = mi + (size_t)f * 2 * C;
        double *fsums = sums + (size_t)f * REP * 3 * C;
        for (int cb = 0; cb < c4; cb += 256) {
            const int col = cb + ct;
            float4 s1 = make_float4(0, 0, 0, 0), s2 = make_float4(0, 0, 0, 0);
            if (rt < rpi && col < c4) {
                const float4 m = *(const float4 *)(fmi + col * 4), iv = *(const float4 *)(fmi + C + col * 4);
                const size_t stride = (size_t)rpi;
                for (size_t r = lo + rt; r < hi; r += TRIP * stride) {
                    float4 g[TRIP], v[TRIP];
#pragma unroll
                    for (int j = 0; j < TRIP; ++j) {
                        const size_t rr = r + j * stride;
                        const bool ok = rr < hi;
                        g[j] = ok ? *(const float4 *)(dyh + rr * C + col * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                        v[j] = ok ? *(const float4 *)(y + rr * C + col * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
#pragma unroll
                    for (int j = 0; j < TRIP; ++j) {
                        s1.x += g[j].x; s1.y += g[j].y; s1.z += g[j].z; s1.w += g[j].w;
                        s2.x += g[j].x * ((v[j].x - m.x) * iv.x); s2.y += g[j].y * ((v[j].y - m.y) * iv.y);
                        s2.z += g[j].z * ((v[j].z - m.z) * iv.z); s2.w += g[j].w * ((v[j].w - m.w) * iv.w);
                    }
                }
            }
            if (64 % c4 == 0) {
                // few channels (c4 divides the wave): the threads of a wave that share a channel quad are folded with shuffles, the
                // four waves through LDS.  (The serial form below walks rpi = 256 / c4 partials per quad with ONE thread per quad:
                // 64 dependent f64 LDS reads x 8 values at C = 16, which made the two 16-channel fusion layers' reduce pass take
                // 94 us for 5 MB.)
                double v[8] = {s1.x, s1.y, s1.z, s1.w, s2.x, s2.y, s2.z, s2.w};
                for (int off = c4; off < 64; off <<= 1)
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] += __shfl_xor(v[i], off, 64);
                const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
                if (lane < c4)
#pragma unroll
                    for (int i = 0; i < 8; ++i) red[i >> 2][wave * c4 + lane][i & 3] = v[i];
                __syncthreads();
                if ((int)threadIdx.x < c4)
                    for (int k = 0; k < 2; ++k)
                        for (int j = 0; j < 4; ++j) {
                            const double t = (red[k][ct][j] + red[k][c4 + ct][j]) + (red[k][2 * c4 + ct][j] + red[k][3 * c4 + ct][j]);
                            atomicAdd(fsums + ((size_t)(blockIdx.x % REP) * 3 + k) * C + col * 4 + j, t);
                        }
                __syncthreads();
            } else {
            red[0][threadIdx.x][0] = s1.x; red[0][threadIdx.x][1] = s1.y; red[0][threadIdx.x][2] = s1.z; red[0][threadIdx.x][3] = s1.w;
            red[1][threadIdx.x][0] = s2.x; red[1][threadIdx.x][1] = s2.y; red[1][threadIdx.x][2] = s2.z; red[1][threadIdx.x][3] = s2.w;
            __syncthreads();
            if (rt == 0 && col < c4) {
                for (int k = 0; k < 2; ++k)
                    for (int j = 0; j < 4; ++j) {
                        double t = 0.0;
                        for (int r = 0; r < rpi; ++r) t += red[k][r * c4 + ct][j];
                        atomicAdd(fsums + ((size_t)(blockIdx.x % REP) * 3 + k) * C + col * 4 + j, t);
                    }
            }
            __syncthreads();
            }
        }
    }
    }
    // ---- the last workgroup: (a, b) of every frame and channel
    mvx_drain_vmem();
    __syncthreads();
    if (threadIdx.x == 0) s_last = (atomicAdd(done_counter, 1u) == gridDim.x - 1u);
    __syncthreads();
    if (!s_last) return;
    for (int e = threadIdx.x; e < C * fm.F; e += blockDim.x) {
        const int f = e / C, c = e - f * C;
        const double *fs = sums + (size_t)f * REP * 3 * C;
        double sa = 0.0, sb = 0.0;
#pragma unroll 1
        for (int r0 = 0; r0 < REP; r0 += 8) {
            double v1[8], v2[8];
#pragma unroll
            for (int rp = 0; rp < 8; ++rp) {
                v1[rp] = __hip_atomic_load(fs + ((size_t)(r0 + rp) * 3 + 0) * C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v2[rp] = __hip_atomic_load(fs + ((size_t)(r0 + rp) * 3 + 1) * C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int rp = 0; rp < 8; ++rp) { sa += v1[rp]; sb += v2[rp]; }
        }
        const double count = fm.count[f];
        ab[(size_t)f * 2 * C + c] = (float)(sa / count);
        ab[(size_t)f * 2 * C + C + c] = (float)(sb / count);
    }
}

// backward, pass 2: dz = (y > 0) ? inv * (dyh - s1/N - yhat * s2/N) : 0 ; dbias[c] += sum dz (over ALL frames)
// PL (compile time): dz is written as THREE PLANES of bf16 pieces (u16 [3][rows][C]: hi + mid + lo = dz exactly,
// split_common.h) instead of f32 -- the operand format of the weight gradient on pre-cut operands (rowgemm_pre.hip), for a layer
// whose dz has no other reader (the first layer of the fusion MLP: its input carries no gradient)
template <int TRIP, bool PL = false>
__global__ __launch_bounds__(256) void bn_bwd_apply(const float *__restrict__ dyh, const float *__restrict__ y,
                                                    const float *__restrict__ mi, const float *__restrict__ ab,
                                                    float *__restrict__ dz, double *__restrict__ dbias,
                                                    const float *__restrict__ row_w, size_t rows, int C,
                                                    unsigned *__restrict__ done_counter, float *__restrict__ dbias_out,
                                                    int accumulate, size_t rows_per_block, FrameMap fm,
                                                    unsigned *__restrict__ amax_slot, unsigned block_base,
                                                    unsigned total_blocks) {
    // block_base / total_blocks: the pass may be enqueued in several launches over consecutive block ranges (the consumer of
    // the first rows starts while the rest is still being written); the bias gradient is folded by the last block of ALL
    __shared__ double red[256][4];
    float mx = 0.f;                                  // max |dz| this thread wrote (-> amax_slot: the fp16-piece kernels scale dz by it)
    const int c4 = C >> 2;
    const int rpi = max(1, 256 / c4);
    const int ct = threadIdx.x % c4, rt = threadIdx.x / c4;
    const unsigned gblock = blockIdx.x + block_base;
    const size_t blk_lo = gblock * rows_per_block;
    const size_t blk_hi = blk_lo + rows_per_block < rows ? blk_lo + rows_per_block : rows;
    const bool live = blk_lo < blk_hi;
    const int s_lo = (fm.F == 1 || !live) ? 0 : fm_seg_of(fm, (long long)blk_lo);
    const int s_hi = (fm.F == 1 || !live) ? 0 : fm_seg_of(fm, (long long)blk_hi - 1);
    for (int cb = 0; cb < c4; cb += 256) {
        const int col = cb + ct;
        float4 sb = make_float4(0, 0, 0, 0);
        for (int sg = s_lo; sg <= s_hi && live; ++sg) {
            const int f = fm.F == 1 ? 0 : (int)fm.seg_frame[sg];
            const size_t lo = fm.F == 1 ? blk_lo : max(blk_lo, (size_t)fm.bound[sg]);
            const size_t hi = fm.F == 1 ? blk_hi : min(blk_hi, (size_t)fm.bound[sg + 1]);
            const float *fmi = mi + (size_t)f * 2 * C;
            if (rt < rpi && col < c4) {
                const float4 m = *(const float4 *)(fmi + col * 4), iv = *(const float4 *)(fmi + C + col * 4);
                // written by the last workgroup of pass 1 (an earlier kernel of this stream): plain loads
                const float4 av = *(const float4 *)(ab + (size_t)f * 2 * C + col * 4);
                const float4 bv = *(const float4 *)(ab + (size_t)f * 2 * C + C + col * 4);
                const float a[4] = {av.x, av.y, av.z, av.w}, b[4] = {bv.x, bv.y, bv.z, bv.w};
                const size_t stride = (size_t)rpi;
                for (size_t r = lo + rt; r < hi; r += TRIP * stride) {
                    float4 g[TRIP], v[TRIP];
                    float rw[TRIP];
#pragma unroll
                    for (int j = 0; j < TRIP; ++j) {
                        const size_t rr = r + j * stride;
                        const bool ok = rr < hi;
                        g[j] = ok ? *(const float4 *)(dyh + rr * C + col * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                        v[j] = ok ? *(const float4 *)(y + rr * C + col * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                        rw[j] = (ok && row_w) ? row_w[rr] : 1.f;   // a compact row standing for rw dense rows
                    }
#pragma unroll
                    for (int j = 0; j < TRIP; ++j) {
                        const size_t rr = r + j * stride;
                        if (rr >= hi) break;
                        float4 o;
                        o.x = v[j].x > 0.f ? iv.x * (g[j].x - rw[j] * (a[0] + ((v[j].x - m.x) * iv.x) * b[0])) : 0.f;
                        o.y = v[j].y > 0.f ? iv.y * (g[j].y - rw[j] * (a[1] + ((v[j].y - m.y) * iv.y) * b[1])) : 0.f;
                        o.z = v[j].z > 0.f ? iv.z * (g[j].z - rw[j] * (a[2] + ((v[j].z - m.z) * iv.z) * b[2])) : 0.f;
                        o.w = v[j].w > 0.f ? iv.w * (g[j].w - rw[j] * (a[3] + ((v[j].w - m.w) * iv.w) * b[3])) : 0.f;
                        if constexpr (PL) {
                            uint2 pc[3];
                            split_n<3, 0>(o.x, o.y, o.z, o.w, pc);
                            unsigned short *pl = (unsigned short *)dz;
#pragma unroll
                            for (int q = 0; q < 3; ++q) *(uint2 *)(pl + ((size_t)q * rows + rr) * C + col * 4) = pc[q];
                        } else {
                            *(float4 *)(dz + rr * C + col * 4) = o;
                        }
                        sb.x += o.x; sb.y += o.y; sb.z += o.z; sb.w += o.w;
                        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
                    }
                }
            }
        }
        if (dbias && 64 % c4 == 0) {                  // see bn_bwd_reduce: shuffles within the wave, four partials through LDS
            double v[4] = {sb.x, sb.y, sb.z, sb.w};
            for (int off = c4; off < 64; off <<= 1)
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] += __shfl_xor(v[i], off, 64);
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            if (lane < c4)
#pragma unroll
                for (int i = 0; i < 4; ++i) red[wave * c4 + lane][i] = v[i];
            __syncthreads();
            if ((int)threadIdx.x < c4)
                for (int j = 0; j < 4; ++j)
                    atomicAdd(dbias + (size_t)(gblock % REP) * 3 * C + col * 4 + j,
                              (red[ct][j] + red[c4 + ct][j]) + (red[2 * c4 + ct][j] + red[3 * c4 + ct][j]));
            __syncthreads();
        } else if (dbias) {
            red[threadIdx.x][0] = sb.x; red[threadIdx.x][1] = sb.y; red[threadIdx.x][2] = sb.z; red[threadIdx.x][3] = sb.w;
            __syncthreads();
            if (rt == 0 && col < c4) {
                for (int j = 0; j < 4; ++j) {
                    double t = 0.0;
                    for (int r = 0; r < rpi; ++r) t += red[r * c4 + ct][j];
                    atomicAdd(dbias + (size_t)(gblock % REP) * 3 * C + col * 4 + j, t);
                }
            }
            __syncthreads();
        }
    }
    if (amax_slot) mvx_wave_amax_to(amax_slot, mx);
    if (dbias && done_counter) {
        // the workgroup that finishes last folds the replicas into the bias gradient (no dbias_finish launch);
        // atomics-only protocol as in bn_finalize_by_last_block (common.h)
        __shared__ int s_last;
        mvx_drain_vmem();
        __syncthreads();
        if (threadIdx.x == 0) s_last = (atomicAdd(done_counter, 1u) == total_blocks - 1u);
        __syncthreads();
        if (s_last) {
            for (int i = threadIdx.x; i < C; i += blockDim.x) {
                double v[REP];
#pragma unroll
                for (int rp = 0; rp < REP; ++rp)
                    v[rp] = __hip_atomic_load(dbias + (size_t)rp * 3 * C + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                double t = 0.0;
#pragma unroll
                for (int rp = 0; rp < REP; ++rp) t += v[rp];
                dbias_out[i] = accumulate ? dbias_out[i] + (float)t : (float)t;
            }
        }
    }
}

// dbias[c] = sum over replicas of scratch[rep][2][c]
__global__ void dbias_finish(const double *__restrict__ scratch, float *__restrict__ b, int C, int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C) return;
    double t = 0.0;
    for (int rp = 0; rp < REP; ++rp) t += scratch[((size_t)rp * 3 + 2) * C + i];
    b[i] = accumulate ? b[i] + (float)t : (float)t;
}

constexpr int BWD_TRIP = 4;            // rows per thread in flight in the two BatchNorm-backward passes

inline unsigned row_grid(size_t rows, int C) {
    const int rpi = (256 / (C / 4)) > 1 ? 256 / (C / 4) : 1;
    size_t b = (rows + 4 * (size_t)rpi - 1) / (4 * (size_t)rpi);   // kernels take 4 rows per thread and trip
    return (unsigned)(b > 768 ? 768 : (b ? b : 1));
}

}  // namespace

extern "C" int mvx_bn_finalize_frames(const double *stats, double count, double eps, float *mean_inv, int32_t channels,
                                      int32_t n_frames, void *stream) {
    MVX_CHECK_ARG(stats && mean_inv && channels > 0 && count > 0 && n_frames >= 1 && n_frames <= MVX_MAX_FRAMES);
    hipLaunchKernelGGL(bn_finalize, dim3(mvx_cdiv(channels, 128), n_frames), dim3(128), 0, (hipStream_t)stream, stats, count,
                       eps, mean_inv, channels);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" int mvx_bn_finalize(const double *stats, double count, double eps, float *mean_inv, int32_t channels,
                               void *stream) {
    return mvx_bn_finalize_frames(stats, count, eps, mean_inv, channels, 1, stream);
}

extern "C" int mvx_bn_apply_frames(const float *y, const float *mean_inv, float *out, int64_t rows, int32_t channels,
                                   const mvx_frames_t *frames_host, int32_t row_kind, void *stream) {
    MVX_CHECK_ARG(y && mean_inv && out && rows >= 0 && channels > 0 && channels % 4 == 0);
    if (rows == 0) return MVX_OK;
    FrameMap fm;
    MVX_CHECK_ARG(mvx_build_frame_map(fm, frames_host, row_kind, rows, 1.0));
    const size_t n4 = (size_t)rows * channels / 4;
    const unsigned grid = (unsigned)(mvx_cdiv(n4, 256) > 4096 ? 4096 : mvx_cdiv(n4, 256));
    hipLaunchKernelGGL(bn_apply, dim3(grid), dim3(256), 0, (hipStream_t)stream, y, mean_inv, out, n4, channels, fm);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" int mvx_bn_apply(const float *y, const float *mean_inv, float *out, int64_t rows, int32_t channels,
                            void *stream) {
    return mvx_bn_apply_frames(y, mean_inv, out, rows, channels, nullptr, MVX_ROWS_SINGLE, stream);
}

extern "C" int mvx_row_stats(const float *y, double *stats, int64_t rows, int32_t channels, void *stream) {
    MVX_CHECK_ARG(y && stats && rows >= 0 && channels > 0 && channels % 4 == 0);
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(stats, 0, sizeof(double) * MVX_REP * 2 * channels, st);
    if (e != hipSuccess) return (int)e;
    if (rows == 0) return MVX_OK;
    hipLaunchKernelGGL(row_stats, dim3(row_grid(rows, channels)), dim3(256), 0, st, y, stats, (size_t)rows, channels);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" size_t mvx_bn_backward_scratch_bytes(int32_t channels) {
    // replicated sums + one slot for the "last workgroup" counter of the fused bias-gradient reduction
    return mvx_bn_backward_scratch_bytes_frames(channels, 1);
}

extern "C" size_t mvx_bn_backward_scratch_bytes_frames(int32_t channels, int32_t n_frames) {
    // per frame: replicated (sum dyh, sum dyh*yhat, dbias replica slot); the bias gradient uses frame 0's third slot
    // ... TWO slots (16 bytes: what follows stays 16-byte aligned for its float4 loads) for the two "last workgroup" counters,
    // and the (a, b) floats of every frame [F][2][C]
    return channels > 0 && n_frames > 0 ? sizeof(double) * ((REP * 3 + 1) * (size_t)channels * n_frames + 2) : 0;
}

// amax[0] = max(amax[0], max |x[i]|) (amax zeroed first unless MVX_FLAG_PREZEROED): operand range of the fp16-piece kernels
// for tensors no kernel of this library produced (sampled image features, the loss gradient)
__global__ __launch_bounds__(256) void tensor_amax(const float *__restrict__ x, size_t n, unsigned *__restrict__ slot) {
    float mx = 0.f;
    const size_t n4 = n >> 2;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = ((const float4 *)x)[i];
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) mx = fmaxf(mx, fabsf(x[(n4 << 2) + threadIdx.x]));
    mvx_wave_amax_to(slot, mx);
}

extern "C" int mvx_tensor_amax(const float *x, int64_t n, float *amax, int32_t flags, void *stream) {
    MVX_CHECK_ARG(x && amax && n >= 0 && ((uintptr_t)x & 15) == 0);
    hipStream_t st = (hipStream_t)stream;
    if (!(flags & MVX_FLAG_PREZEROED)) {
        hipError_t e = hipMemsetAsync(amax, 0, sizeof(float), st);
        if (e != hipSuccess) return (int)e;
    }
    if (n == 0) return MVX_OK;
    const size_t want = ((size_t)n / 4 + 255) / 256;
    hipLaunchKernelGGL(tensor_amax, dim3((unsigned)(want > 2048 ? 2048 : (want ? want : 1))), dim3(256), 0, st, x, (size_t)n,
                       (unsigned *)amax);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

// fp16 pieces: a weight is cut times SPLIT_F16_WSCALE = 2^8, so |w| must stay below 65504 / 256 (split_common.h).  One status bit
// says that a tensor does not: the guard of the fp16x3 arithmetic (modules/_hip.py guard_fp16_weight), checked with the other
// data-dependent status words once per step.
__global__ __launch_bounds__(256) void f16_weight_check(const float *__restrict__ w, size_t n, int *__restrict__ status) {
    bool bad = false;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        const float v = fabsf(w[e]);
        bad |= !(v < 65504.f / SPLIT_F16_WSCALE);            // also true for NaN
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(status, MVX_STATUS_F16_WEIGHT_RANGE);
}

extern "C" int mvx_split_f16_weight_check(const float *w, int64_t n, int32_t *status, void *stream) {
    MVX_CHECK_ARG(w && status && n >= 0);
    if (n == 0) return MVX_OK;
    const size_t want = ((size_t)n + 255) / 256;
    hipLaunchKernelGGL(f16_weight_check, dim3((unsigned)(want > 1024 ? 1024 : want)), dim3(256), 0, (hipStream_t)stream, w, (size_t)n,
                       status);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

static int bn_relu_backward_impl(const float *dyhat, const float *y, const float *mean_inv, double count,
                                 float *dz, float *dbias, double *scratch, const float *row_w, int64_t rows,
                                 int32_t channels, int32_t flags, const mvx_frames_t *frames_host,
                                 int32_t row_kind, float *dz_amax, void *stream, bool planes, int part = 0, int nparts = 1,
                                 int64_t *part_rows = nullptr) {
    MVX_CHECK_ARG(dyhat && y && mean_inv && dz && scratch && rows >= 0 && channels > 0 && channels % 4 == 0);
    MVX_CHECK_ARG(nparts >= 1 && part >= 0 && part < nparts);
    MVX_CHECK_ARG(count > 0);
    MVX_CHECK_ARG(((uintptr_t)scratch & 15) == 0);          // bn_bwd_apply reads the (a, b) floats behind the sums with float4 loads
    FrameMap fm;
    MVX_CHECK_ARG(mvx_build_frame_map(fm, frames_host, row_kind, rows, count));
    hipStream_t st = (hipStream_t)stream;
    const size_t slots = (size_t)REP * 3 * channels * fm.F;
    if (!(flags & MVX_FLAG_PREZEROED) && part == 0) {
        hipError_t e = hipMemsetAsync(scratch, 0, sizeof(double) * (slots + 2), st);
        if (e != hipSuccess) return (int)e;
    }
    if (part_rows) part_rows[0] = part_rows[1] = 0;
    if (rows > 0) {
        const int rpi = (256 / (channels / 4)) > 1 ? 256 / (channels / 4) : 1;
        // up to 4 workgroups per CU, each with at least two trips of BWD_TRIP x rpi rows
        size_t want = ((size_t)rows + 2 * BWD_TRIP * rpi - 1) / (2 * (size_t)BWD_TRIP * rpi);
        const unsigned grid = (unsigned)(want > 1024 ? 1024 : (want ? want : 1));
        size_t rpb = ((size_t)rows + grid - 1) / grid;
        rpb = (rpb + BWD_TRIP * rpi - 1) / (BWD_TRIP * (size_t)rpi) * (BWD_TRIP * (size_t)rpi);       // whole (TRIP x rpi)-row trips
        const unsigned blocks = (unsigned)(((size_t)rows + rpb - 1) / rpb);
        unsigned *counters = (unsigned *)(scratch + slots);           // [0] pass 2 (bias gradient), [1] pass 1 ((a, b) finalisation)
        float *ab = (float *)(scratch + slots + 2);
        // the reduction pass covers all rows and belongs to part 0; the apply pass of part p covers the blocks
        // [blocks p / nparts, blocks (p + 1) / nparts) -- whole blocks, so the parts' row ranges tile [0, rows)
        const unsigned b_lo = (unsigned)((size_t)blocks * part / nparts), b_hi = (unsigned)((size_t)blocks * (part + 1) / nparts);
        if (part_rows) {
            part_rows[0] = (int64_t)((size_t)b_lo * rpb);
            part_rows[1] = (int64_t)((size_t)b_hi * rpb < (size_t)rows ? (size_t)b_hi * rpb : (size_t)rows);
        }
        if (part == 0) {
            hipLaunchKernelGGL(bn_bwd_reduce<BWD_TRIP>, dim3(blocks), dim3(256), 0, st, dyhat, y, mean_inv, scratch, (size_t)rows,
                               channels, rpb, fm, counters + 1, ab, (unsigned *)dz_amax);
            MVX_LAUNCH_CHECK();
        }
        if (b_hi > b_lo) {
            if (planes)
                hipLaunchKernelGGL((bn_bwd_apply<BWD_TRIP, true>), dim3(b_hi - b_lo), dim3(256), 0, st, dyhat, y, mean_inv,
                                   (const float *)ab, dz, dbias ? scratch + 2 * channels : (double *)nullptr, row_w, (size_t)rows,
                                   channels, counters, dbias, flags & MVX_FLAG_ACCUMULATE, rpb, fm, (unsigned *)dz_amax, b_lo, blocks);
            else
                hipLaunchKernelGGL((bn_bwd_apply<BWD_TRIP, false>), dim3(b_hi - b_lo), dim3(256), 0, st, dyhat, y, mean_inv,
                                   (const float *)ab, dz, dbias ? scratch + 2 * channels : (double *)nullptr, row_w, (size_t)rows,
                                   channels, counters, dbias, flags & MVX_FLAG_ACCUMULATE, rpb, fm, (unsigned *)dz_amax, b_lo, blocks);
            MVX_LAUNCH_CHECK();
        }
    } else if (dz_amax) {
        hipError_t e = hipMemsetAsync(dz_amax, 0, sizeof(float), st);
        if (e != hipSuccess) return (int)e;
    }
    if (rows <= 0 && dbias && part == 0) {          // no rows: the bias gradient is zero
        hipLaunchKernelGGL(dbias_finish, dim3(mvx_cdiv(channels, 128)), dim3(128), 0, st, (const double *)scratch, dbias,
                           channels, flags & MVX_FLAG_ACCUMULATE);
        MVX_LAUNCH_CHECK();
    }
    return MVX_OK;
}

extern "C" int mvx_bn_relu_backward_frames(const float *dyhat, const float *y, const float *mean_inv, double count,
                                           float *dz, float *dbias, double *scratch, const float *row_w, int64_t rows,
                                           int32_t channels, int32_t flags, const mvx_frames_t *frames_host,
                                           int32_t row_kind, float *dz_amax, void *stream) {
    return bn_relu_backward_impl(dyhat, y, mean_inv, count, dz, dbias, scratch, row_w, rows, channels, flags, frames_host, row_kind,
                                 dz_amax, stream, false);
}

// The same with dz written as three planes of bf16 pieces, u16 [3][rows][channels] (MVX_FLAG_SPLIT3 operand format of
// mvx_linear_wgrad_pre), instead of f32: for a layer whose dz only feeds its own weight gradient.
extern "C" int mvx_bn_relu_backward_planes_frames(const float *dyhat, const float *y, const float *mean_inv, double count,
                                                  void *dz_planes, float *dbias, double *scratch, const float *row_w, int64_t rows,
                                                  int32_t channels, int32_t flags, const mvx_frames_t *frames_host,
                                                  int32_t row_kind, float *dz_amax, void *stream) {
    MVX_CHECK_ARG((((uintptr_t)dz_planes) & 15) == 0);
    return bn_relu_backward_impl(dyhat, y, mean_inv, count, (float *)dz_planes, dbias, scratch, row_w, rows, channels, flags,
                                 frames_host, row_kind, dz_amax, stream, true);
}

// ... enqueued in `nparts` calls (part = 0 .. nparts - 1, in this order, same arguments, same stream): part 0 runs the reduction
// pass over all rows and the apply pass of the first row range, part p > 0 the apply pass of its range; part_rows[0..1] receives
// the range [lo, hi) of rows whose planes THIS call writes (the ranges tile [0, rows)).  The weight gradient of the rows of part p
// (mvx_linear_wgrad_pre_rows) can then run beside the apply pass of part p + 1.  The bias gradient is complete after the last part.
extern "C" int mvx_bn_relu_backward_planes_part_frames(const float *dyhat, const float *y, const float *mean_inv, double count,
                                                       void *dz_planes, float *dbias, double *scratch, const float *row_w,
                                                       int64_t rows, int32_t channels, int32_t flags,
                                                       const mvx_frames_t *frames_host, int32_t row_kind, int32_t part,
                                                       int32_t nparts, int64_t *part_rows, void *stream) {
    MVX_CHECK_ARG((((uintptr_t)dz_planes) & 15) == 0 && part_rows);
    return bn_relu_backward_impl(dyhat, y, mean_inv, count, (float *)dz_planes, dbias, scratch, row_w, rows, channels, flags,
                                 frames_host, row_kind, nullptr, stream, true, part, nparts, part_rows);
}

extern "C" int mvx_bn_relu_backward(const float *dyhat, const float *y, const float *mean_inv, double count,
                                    float *dz, float *dbias, double *scratch, const float *row_w, int64_t rows,
                                    int32_t channels, int32_t flags, void *stream) {
    return mvx_bn_relu_backward_frames(dyhat, y, mean_inv, count, dz, dbias, scratch, row_w, rows, channels, flags, nullptr,
                                       MVX_ROWS_SINGLE, nullptr, stream);
}
