// Shared device/host helpers for libmvx_hip (gfx950 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mvx_hip.h"

#define MVX_WAVE 64
#define MVX_REP MVX_STATS_REPLICAS

// A call rejected by its argument checks launches nothing -- and must not leave the operand ranges that
// mvx_split_operand_amax bound for it pending for the NEXT split launch of this thread (a forward convolution would then
// scale its activations by a gradient's 2^27: ADVICE r04).  Every failed check drops the binding.
void mvxi_drop_split_amax();
#define MVX_CHECK_ARG(cond)             \
    do {                                \
        if (!(cond)) {                  \
            mvxi_drop_split_amax();     \
            return MVX_EINVAL;          \
        }                               \
    } while (0)

void mvxi_gather_narrow_max_units(long long v);      // csrc/conv3d.hip: tuning value MVX_TUNE_GATHER_NARROW_MAX_UNITS
void mvxi_count_launch();       // diagnostics only: kernel launches issued by the library (mvx_launch_count)
#define MVX_LAUNCH_CHECK()                              \
    do {                                                \
        mvxi_count_launch();                            \
        hipError_t e__ = hipGetLastError();             \
        if (e__ != hipSuccess) return (int)e__;         \
    } while (0)

static inline unsigned mvx_cdiv(long long a, long long b) { return (unsigned)((a + b - 1) / b); }

// ---- internal helpers shared between translation units (NOT part of the C ABI) -----------------------
// conv3d.hip: compacted (plane, tile) step lists of the background-aware weight gradient and its closed-form term
struct FrameMap;
// fp16-piece operand scaling (split_common.h): the addresses bound by mvx_split_operand_amax for the calling thread's next split
// launch; taking them clears the binding (defined in voxelize.hip)
struct SplitAmax {
    const float *a, *b;      // max |value| of the first / second f32 operand of the launch (device addresses), or NULL
    int coarse_a;            // operand a is a FORWARD operand from outside the library: coarse scale (split_scale_coarse)
};
SplitAmax mvxi_take_split_amax();

// arithmetic code of the split kernels from a flags word: 2 = bf16x3, 3 = bf16x6, 4 = fp16x3 (two fp16 pieces)
static inline int mvx_split_code(int flags) { return (flags & MVX_FLAG_SPLIT_F16) ? 4 : (flags & MVX_FLAG_SPLIT3) ? 3 : 2; }

int mvxi_linear_forward_split(const float *x, int ldx, const float *w, int ldw, const float *bias, float *y,
                              int ldy, double *stats, const float *row_w, long long rows, int k, int n, int relu,
                              unsigned *fin_counter, double fin_eps, float *fin_mean_inv, const FrameMap &fm, int pieces,
                              hipStream_t st, const SplitAmax &am = SplitAmax{nullptr, nullptr, 0});
// K = 128 (rowgemm_k128.hip): weights resident in LDS, rows streamed through registers; same contract and numbers
bool mvxi_rowgemm_k128_ok(int ldx, int ldw, int ldy, int k, int n);
void mvxi_rowgemm_k128_enable(long long v);
int mvxi_linear_forward_k128(const float *x, int ldx, const float *w, int ldw, const float *bias, float *y, int ldy, double *stats,
                             const float *row_w, long long rows, int n, int relu, unsigned *fin_counter, double fin_eps,
                             float *fin_mean_inv, const FrameMap &fm, int pieces, hipStream_t st, const SplitAmax &am);
int mvxi_linear_wgrad_split(const float *x, int ldx, const float *dz, int lddz, float *slabs, long long rows, int k, int n,
                            long long rows_per_strip, long long strips, int pieces, hipStream_t st,
                            const SplitAmax &am = SplitAmax{nullptr, nullptr, 0});
int mvxi_wgrad_step_list(const int32_t *in_halo_flags, int din, int dout, int ntiles, int stride_d, int pad_d, int *list,
                         int *count, hipStream_t st, int n_frames = 1);
int mvxi_wgrad_rank1(const float *tap_sums, const float *c_in, float *dw, int din, int dout, int cin, int cout, int stride_d,
                     int pad_d, hipStream_t st, int n_frames = 1);

// ---- frame sets: the frames of a step processed by ONE launch -----------------------------------------
// The reference is strictly batch-1 (config.yml:18, VoxelNet.py:19): a batch is B independent forwards with per-frame
// BatchNorm statistics.  Every kernel that reduces over "the batch" therefore needs to know which frame a row (or a
// depth plane) belongs to.  Row matrices hold the frames back to back as up to 2F segments (see mvx_frames_t in the
// header); grids stack the frames along the depth axis: global plane = frame * planes_per_frame + local plane.
#define MVX_MAX_SEGS (2 * MVX_MAX_FRAMES)
struct FrameMap {
    int F, nseg;
    int bound[MVX_MAX_SEGS + 1];                // segment s = rows [bound[s], bound[s+1])
    unsigned char seg_frame[MVX_MAX_SEGS];
    double count[MVX_MAX_FRAMES];               // BatchNorm population of frame f (rows of the DENSE tensor it stands for)
};
__device__ __forceinline__ int fm_seg_of(const FrameMap &m, long long r) {
    int s = 0;
    for (int k = 1; k < m.nseg; ++k) s += (r >= (long long)m.bound[k]);
    return s;
}
__device__ __forceinline__ int fm_frame_of(const FrameMap &m, long long r) { return m.F == 1 ? 0 : (int)m.seg_frame[fm_seg_of(m, r)]; }

// Host side: segment table of a row layout.  kind: MVX_ROWS_* of the header; fr == NULL -> one frame of `rows` rows
// with population `count`.  Returns false on an inconsistent description.
static inline bool mvx_build_frame_map(FrameMap &m, const mvx_frames_t *fr, int kind, long long rows, double count) {
    for (int k = 0; k <= MVX_MAX_SEGS; ++k) m.bound[k] = 0;
    for (int k = 0; k < MVX_MAX_SEGS; ++k) m.seg_frame[k] = 0;
    for (int k = 0; k < MVX_MAX_FRAMES; ++k) m.count[k] = 1.0;
    if (!fr || kind == MVX_ROWS_SINGLE) {
        m.F = 1; m.nseg = 1; m.bound[0] = 0; m.bound[1] = (int)rows; m.count[0] = count;
        return rows < (1ll << 31);
    }
    const int F = fr->n_frames;
    if (F < 1 || F > MVX_MAX_FRAMES || fr->t < 1) return false;
    m.F = F;
    for (int f = 0; f < F; ++f) {
        if (fr->real_off[f + 1] < fr->real_off[f] || fr->vox_off[f + 1] < fr->vox_off[f]) return false;
        m.count[f] = (double)(fr->vox_off[f + 1] - fr->vox_off[f]) * (double)fr->t;
    }
    const int R = fr->real_off[F], V = fr->vox_off[F];
    int s = 0;
    if (kind == MVX_ROWS_FUSION || kind == MVX_ROWS_VFE) {
        for (int f = 0; f < F; ++f) { m.bound[s] = fr->real_off[f]; m.seg_frame[s++] = (unsigned char)f; }
        for (int f = 0; f < F; ++f) {
            m.bound[s] = R + (kind == MVX_ROWS_VFE ? fr->vox_off[f] : f);
            m.seg_frame[s++] = (unsigned char)f;
        }
        m.bound[s] = R + (kind == MVX_ROWS_VFE ? V : F);
    } else if (kind == MVX_ROWS_VOXELS) {
        for (int f = 0; f < F; ++f) { m.bound[s] = fr->vox_off[f]; m.seg_frame[s++] = (unsigned char)f; }
        m.bound[s] = V;
    } else if (kind == MVX_ROWS_REAL) {
        for (int f = 0; f < F; ++f) { m.bound[s] = fr->real_off[f]; m.seg_frame[s++] = (unsigned char)f; }
        m.bound[s] = R;
    } else if (kind == MVX_ROWS_GRID) {
        if (rows % F) return false;
        const long long per = rows / F;
        if (rows >= (1ll << 31)) return false;
        for (int f = 0; f < F; ++f) { m.bound[s] = (int)(per * f); m.seg_frame[s++] = (unsigned char)f; m.count[f] = (double)per; }
        m.bound[s] = (int)rows;
    } else {
        return false;
    }
    m.nseg = s;
    return (long long)m.bound[s] == rows;
}

// Frames stacked along the depth axis of a grid: source plane (global) of global output plane d for depth tap kd of a
// convolution with per-frame depths din -> dout, or -1 when the tap falls into the padding.
__device__ __forceinline__ int mvx_src_plane(int d, int din, int dout, int sd, int pd, int kd) {
    const int f = d / dout, dl = d - f * dout;
    const int s = dl * sd - pd + kd;
    return (s >= 0 && s < din) ? f * din + s : -1;
}
// ... and the global OUTPUT plane that reads global input plane d through depth tap kd, or -1
__device__ __forceinline__ int mvx_dst_plane(int d, int din, int dout, int sd, int pd, int kd) {
    const int f = d / din, dl = d - f * din;
    const int t = dl + pd - kd;
    if (t < 0 || (t % sd) != 0) return -1;
    const int o = t / sd;
    return o < dout ? f * dout + o : -1;
}

// ---- wave / block reductions and scans -------------------------------------------------
__device__ __forceinline__ int wave_incl_scan_i32(int v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// Exclusive scan across a 1024-thread block (16 waves).  `smem` needs 17 ints.  Returns the
// exclusive prefix of this thread's value; *total receives the block sum.
__device__ __forceinline__ int block_excl_scan_i32(int v, int *smem, int *total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int inc = wave_incl_scan_i32(v);
    if (lane == 63) smem[wid] = inc;
    __syncthreads();
    if (wid == 0) {
        int w = lane < nw ? smem[lane] : 0;
        int wi = wave_incl_scan_i32(w);
        if (lane < nw) smem[lane] = wi - w;
        if (lane == nw - 1) smem[16] = wi;
    }
    __syncthreads();
    int res = inc - v + smem[wid];
    *total = smem[16];
    __syncthreads();
    return res;
}

// ---- BatchNorm finalisation by the last workgroup of the producing kernel ----------------------------
// Every workgroup calls this after its statistics atomics (all threads, block-uniform arguments).  The workgroup that
// arrives last turns the replicated sums into mean and 1/sqrt(var + eps), which saves the separate mvx_bn_finalize
// launch.  Only device-scope ATOMICS carry the protocol (sums, counter, and the final reads), all served by the same
// coherence point, so no agent-scope fence is needed: such a fence writes the XCD's L2 back (the output tile each
// workgroup has just stored) and cost 13 % of the step when it was tried.  What the protocol does need is that every
// thread's statistics atomics have been PERFORMED at L2 before thread 0 bumps the counter: a workgroup-scope release
// emits no wait on gfx950 (the round-1 binary had the non-returning global_atomic_add_f64 followed by s_barrier and the
// counter atomic with nothing in between), so each thread drains its own vector-memory counter explicitly
// (mvx_drain_vmem: `s_waitcnt vmcnt(0)`; non-returning atomics are vmcnt-tracked on gfx9 and acknowledged by L2 once
// performed) before the barrier.  No cache write-back is involved.  `s_flag` is one int of LDS.
__device__ __forceinline__ void mvx_drain_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// max |value| of a tensor for the fp16-piece kernels (split_common.h): every thread brings the largest magnitude it wrote, one
// atomic per wave folds it into the slot (a zeroed unsigned: the bit patterns of non-negative floats order like the floats)
__device__ __forceinline__ void mvx_wave_amax_to(unsigned *slot, float mx) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    // thousands of waves aim at ONE address: look first (a relaxed load of the device-coherent value) and only send the atomic
    // when it would raise the slot -- after the first few waves almost none does (the unconditional atomics cost 23 us per
    // BatchNorm-backward launch and held mvx_tensor_amax to 1.6 TB/s)
    if ((threadIdx.x & 63) == 0 && mx > 0.f) {
        const unsigned bits = __float_as_uint(mx);
        if (bits > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, bits);
    }
}

template <typename CountOf>
__device__ __forceinline__ void bn_finalize_core(unsigned *done_counter, unsigned total_blocks, double *stats, int C, int F,
                                                 CountOf count_of, double eps, float *mean_inv, int *s_flag) {
    mvx_drain_vmem();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned prev = atomicAdd(done_counter, 1u);
        *s_flag = (prev == total_blocks - 1u);
    }
    __syncthreads();
    if (!*s_flag) return;
    // every frame of the launch: stats [F][REP][2][C] -> mean_inv [F][2][C]
    for (int e = threadIdx.x; e < C * F; e += blockDim.x) {
        const int f = e / C, c = e - f * C;
        const double *st = stats + (size_t)f * MVX_REP * 2 * C;
        // 16 independent device-scope reads in flight per round: the registers of this (one workgroup per launch) tail
        // are allocated for every wave of the kernel -- 64 reads in flight cost the row GEMM 128 VGPRs and with them its
        // second wave per SIMD
        constexpr int CH = 8;
        double s1 = 0.0, s2 = 0.0;
#pragma unroll 1
        for (int r0 = 0; r0 < MVX_REP; r0 += CH) {
            double v1[CH], v2[CH];
#pragma unroll
            for (int rp = 0; rp < CH; ++rp) {
                v1[rp] = __hip_atomic_load(st + ((size_t)(r0 + rp) * 2) * C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v2[rp] = __hip_atomic_load(st + ((size_t)(r0 + rp) * 2 + 1) * C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int rp = 0; rp < CH; ++rp) { s1 += v1[rp]; s2 += v2[rp]; }
        }
        const double count = count_of(f);
        const double mean = s1 / count;
        double var = s2 / count - mean * mean;
        if (var < 0.0) var = 0.0;
        mean_inv[(size_t)f * 2 * C + c] = (float)mean;
        mean_inv[(size_t)f * 2 * C + C + c] = (float)(1.0 / sqrt(var + eps));
    }
}

// row matrices: per-frame populations from the frame map
__device__ __forceinline__ void bn_finalize_by_last_block(unsigned *done_counter, unsigned total_blocks,
                                                          double *stats, int C, const FrameMap &fm, double eps,
                                                          float *mean_inv, int *s_flag) {
    bn_finalize_core(done_counter, total_blocks, stats, C, fm.F, [&](int f) { return fm.count[f]; }, eps, mean_inv, s_flag);
}

// grids / single frames: the same population for every frame of the launch
__device__ __forceinline__ void bn_finalize_by_last_block(unsigned *done_counter, unsigned total_blocks,
                                                          double *stats, int C, double count, double eps,
                                                          float *mean_inv, int *s_flag, int n_frames = 1) {
    bn_finalize_core(done_counter, total_blocks, stats, C, n_frames, [=](int) { return count; }, eps, mean_inv, s_flag);
}

__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
    return v;
}
