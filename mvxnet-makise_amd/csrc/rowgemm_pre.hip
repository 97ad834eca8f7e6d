// Row GEMMs on PRE-CUT operands: the wide layers of the fusion MLP (imhead/Pipe.py:84-104: nn.Linear / 1x1 Conv2d of
// modules/layers/Blocks.py:5-18,31-40) in the split arithmetics of split_common.h, with the cut taken OUT of the GEMM and the
// tile sized for what a CU can be FED.
//
// What bounds linear_split.hip (128 x 128 tiles, f32 operands cut in registers, two workgroups per CU) was measured this round
// with in-kernel s_memtime stamps and counters on a DMA-fed 256 x 128 version of the same product:
// the loop was issue-lean (2 other instructions per MFMA) and still ran at 0.45 of the matrix pipe, in the SAME time as the
// old kernel, because a CU takes in only ~12 bytes per clock from L2 / fabric at this hit mix (72 KB per stage arrived 4,300
// cycles after issue) while the matrix pipe at full rate wants  bytes-per-MFMA x 1/8 MFMA per clock:  187 B per MFMA for a
// 256 x 128 tile of three 16-bit planes -> at most 0.51 busy.  The lever is bytes per MFMA, i.e. the tile:
//   * workgroup tile 256 x 256 (8 waves as 2 x 4, wave tile 128 x 64 = 4 x 2 MFMA tiles of 32 x 32, 128 accumulator
//     registers): 125 B per MFMA with three planes, 83 with two;
//   * operands live in HBM as NP planes of 16-bit pieces ([piece][row][k], written that way by their producers -- the sampler,
//     BatchNorm apply / backward, a per-step weight pack) and travel global -> LDS by `global_load_lds_dwordx4` (no VGPRs, no
//     ds_write, no conversion in the GEMM);
//   * K in steps of 16 (one MFMA k-step): a stage is NP x 512 rows x 32 B = 48 KB (NP = 3), three stages = 144 KB of the CU's
//     160 KB: one workgroup per CU, two waves per SIMD;
//   * the two waves of a SIMD run in ANTI-PHASE (waves w and w + 4 share a SIMD): time is cut into intervals by workgroup
//     barriers; group 0 = waves 0-3 runs  LOAD(u) | MFMA(u)  for k-step u = 0, 1, ..., group 1 the same one interval later, so
//     in every interval one wave of each SIMD issues its 48 MFMAs while the other reads the 18 fragments of its next k-step
//     (and, group 1 only, issues the DMA of the stage two k-steps ahead: an LDS-DMA instruction costs ~75 cycles to issue and
//     must not sit in front of MFMAs).  In lockstep both waves wait for LDS, for the DMA and at the barrier together and the
//     pipe idles (measured: 0.31 of the wave cycles parked).
// Hazards (cdna_hip_programming.md section 5, "read a staged buffer one phase after the wait that retires it"): stage v uses
// buffer v % 3; group 0 reads it in interval 2v, group 1 in 2v + 1, their lgkmcnt(0) sits behind the next barrier, so the
// buffer is free from interval 2v + 3 on -- exactly where group 1 issues stage v + 3 (its LOAD(v + 1) = interval 2v + 3).
// Group 1 drains its DMA of stage v (counted vmcnt: the younger stage stays in flight) at the end of interval 2v - 1; the
// barrier into interval 2v publishes it.
//
// x = the sum of its pieces EXACTLY for NP = 3 bf16 pieces (bf16x6) and products accumulate k-step by k-step in ascending k
// with the piece order of split_mac2: the forward is bit-identical to linear_fwd_split.
#include "common.h"
#include "split_common.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
constexpr int PT = 256;                 // workgroup tile: PT x PT
constexpr int PKS = 16;                 // k-step = stage depth
constexpr int NSTG = 3;

// ---- f32 rows -> NP planes of 16-bit pieces (weights once per optimizer step; any tensor whose producer does not write planes)
template <int NP, int FMT>
__global__ __launch_bounds__(256) void split_rows_kernel(const float *__restrict__ x, int ldx, long long rows, int K,
                                                         unsigned short *__restrict__ planes, long long plane_stride, int ldp,
                                                         float scale) {
    const int kq = K >> 2;
    const long long total = rows * kq;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long r = e / kq;
        const int part = (int)(e - r * kq);
        f32x4 v = *(const f32x4 *)(x + r * ldx + part * 4);
        if constexpr (FMT == 1) v *= scale;
        uint2 pc[NP];
        split_n<NP, FMT>(v[0], v[1], v[2], v[3], pc);
#pragma unroll
        for (int q = 0; q < NP; ++q) *(uint2 *)(planes + q * plane_stride + r * ldp + part * 4) = pc[q];
    }
}

__device__ __forceinline__ bf16x8 pre_tr_frag(const unsigned char *p) {
    typedef __attribute__((address_space(3))) s16x4 lds4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4 *)p);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4 *)(p + 256));      // four rows further
    s16x8 v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
    v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(bf16x8, v);
}

// The shared main loop.  WG = 0: C[i][j] += sum_k A[i][k] B[j][k]  (forward / input gradient; planes [piece][row][k], a stage
// holds 16 k of 256 rows of each operand as 32-byte rows; slot s of row r sits at physical slot s ^ ((r >> 4) & 1) so that
// the ds_read_b128 lane groups of a 32-row fragment hit 16 distinct 16-byte bank groups; a DMA instruction writes 1 KB = 32
// rows lane-linearly, so the swizzle goes on the SOURCE address).  WG = 1: C[i][j] += sum_r A[r][i] B[r][j] (weight gradient;
// planes [piece][row][column], a stage holds 16 rows of 256 columns of each operand as [32-column block][row][32] = 64-byte
// rows -- a DMA instruction writes one column block -- fetched with ds_read_b64_tr_b16, whose lane groups cover four
// consecutive 64-byte rows: no swizzle needed).
// `src[i]`: byte offset (from the operand's plane base) of what this lane loads for DMA chunk (wv & 3) + 4 i of stage 0; stage v
// adds v * astep / v * bstep bytes, except the LAST stage, whose offsets are `src_last[i]` (weight gradient: rows clamped into
// the tensor; the caller passes `tail_rows` < 16 and the rows from there on are zeroed in LDS).
template <int NP, int FMT, int WG, bool STAMP>
__device__ __forceinline__ void pre_mainloop(unsigned char *smem, f32x16 (&acc)[4][2], const unsigned char *__restrict__ a,
                                             const unsigned char *__restrict__ b, const unsigned (&src)[NP * 2],
                                             const unsigned (&src_last)[NP * 2], unsigned astep, unsigned bstep, int U, int wv,
                                             int lane, int tail_rows, unsigned long long *stamps, bool do_stamp) {
    constexpr int CH = NP * 16;                        // 1-KB DMA chunks per stage: A pieces x 8, then B pieces x 8
    constexpr int NCH = CH / 8;                        // per wave: group 1's waves move the A half of a stage, group 0's the B half
    constexpr int STAGE = CH * 1024;
    constexpr int HALF = NP * 8 * 1024;                // byte offset of the B tiles inside a stage
    const int li = lane & 31, lh = lane >> 5;
    // wave tile: rows tm * 128 ..., columns tn * 64 ...; group g = wv >> 2 = tm: the two groups own the upper / lower 128 rows
    // of the tile, and waves w and w + 4 (same SIMD) are in different groups, which is what the anti-phase needs
    const int g = wv >> 2, tm = wv >> 2, tn = wv & 3;
    int n_stamp = 0;
    auto stamp = [&]() __attribute__((always_inline)) {
        if constexpr (STAMP) {
            if (do_stamp && lane == 0 && n_stamp < 512) stamps[wv * 512 + n_stamp] = __builtin_amdgcn_s_memtime();
            ++n_stamp;
        }
    };
    auto bar = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    // DMA of this wave's share of stage v into buffer v % 3: chunk (wv & 3) + 4 i of the A half (group 1) / the B half (group 0)
    const unsigned char *mine = g == 1 ? a : b;
    const unsigned mystep = g == 1 ? astep : bstep;
    const int myhalf = g == 1 ? 0 : HALF;
    auto issue = [&](int v) __attribute__((always_inline)) {
        unsigned char *dst = smem + (v % NSTG) * STAGE + myhalf;
        const bool last = v == U - 1;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = (wv & 3) + 4 * i;
            const unsigned off = last ? src_last[i] : src[i] + (unsigned)v * mystep;
            __builtin_amdgcn_global_load_lds((const void *)(mine + off), (lds_void *)(dst + c * 1024), 16, 0, 0);
        }
    };
    // fragment addresses inside a stage
    int fa, fb;
    if constexpr (WG == 0) {
        const int lo = li * 32 + (((lh ^ (li >> 4)) & 1) << 4);
        fa = tm * 128 * 32 + lo;
        fb = HALF + tn * 64 * 32 + lo;
    } else {
        const int grp = lane >> 4, i16 = lane & 15;
        const int lo = ((grp >> 1) * 8 + (i16 >> 2)) * 64 + ((grp & 1) * 16 + 4 * (i16 & 3)) * 2;
        fa = tm * 4 * 1024 + lo;
        fb = HALF + tn * 2 * 1024 + lo;
    }

    // The DMA is shared by the two LOAD phases (an LDS-DMA instruction costs 75-150 cycles to issue: with all twelve per
    // k-step in group 1 its LOAD phase took 2,200 cycles against 1,660 of MFMAs, measured with the stamps): group 1 moves the A
    // half of stage u + 2 in its LOAD(u) (interval 2u + 1: three intervals before the stage's first read), group 0 the B half of
    // stage u + 1 in its LOAD(u) (interval 2u: two intervals before); buffer (u + 1) % 3 was last read in interval 2u - 3.
    if (g == 1) {
        issue(0);
        if (U > 1) {
            issue(1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NCH) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    } else {
        issue(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    bar();
    if (g == 1) bar();                                             // one interval behind
    for (int u = 0; u < U; ++u) {
        unsigned char *sb = smem + (u % NSTG) * STAGE;
        if (WG == 1 && u == U - 1 && tail_rows < PKS) {
            // partial last stage of a weight-gradient strip: rows >= tail_rows are not part of the strip (the DMA read clamped
            // rows): zero them.  Every wave writes ALL those rows (the same zeros from eight waves), so each wave's reads below
            // see its own writes -- no cross-wave ordering needed; the stage has landed (barrier into this group's LOAD interval)
            for (int e = lane; e < STAGE / 16; e += 64)
                if (((e >> 2) & 15) >= tail_rows) *(uint4 *)(sb + e * 16) = uint4{0u, 0u, 0u, 0u};
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        bf16x8 av[4][NP], bv[2][NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if constexpr (WG == 0) av[t][q] = __builtin_bit_cast(bf16x8, *(const uint4 *)(sb + fa + q * 8192 + t * 1024));
                else av[t][q] = pre_tr_frag(sb + fa + q * 8192 + t * 1024);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if constexpr (WG == 0) bv[t][q] = __builtin_bit_cast(bf16x8, *(const uint4 *)(sb + fb + q * 8192 + t * 1024));
                else bv[t][q] = pre_tr_frag(sb + fb + q * 8192 + t * 1024);
            }
        }
        stamp();                                                       // 0: reads issued
        if (g == 1) {
            if (u + 2 < U) {
                issue(u + 2);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NCH) : "memory");      // A of stage u + 1 has landed; u + 2 stays in flight
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        } else if (u + 1 < U) {
            issue(u + 1);                                              // B of stage u + 1: drained behind this group's MFMAs
        }
        stamp();                                                       // 1: DMA issued (group 1: older stage drained)
        bar();
        stamp();                                                       // 2: through the barrier
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        stamp();                                                       // 3: fragments in registers
#pragma unroll
        for (int m = 0; m < 4; ++m) split_mac2<NP, FMT>(acc[m][0], acc[m][1], av[m], bv[0], bv[1]);
        __builtin_amdgcn_sched_barrier(0);
        if (g == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // B of stage u + 1 has landed (issued in this group's LOAD(u))
        stamp();                                                       // 4: 48 MFMAs issued
        bar();
        stamp();                                                       // 5: through the barrier
    }
    if (g == 0) bar();                                                 // group 0 joins group 1's last barrier
}

// XCD-aware tile order: XCD x = h % 8 runs the row blocks {8 j + x} and a row block's column blocks in consecutive slots, so
// the blocks that share rows share an L2 (speed only).
struct PreTile { unsigned rb, cb; bool on; };
__device__ __forceinline__ PreTile pre_tile(unsigned h, unsigned nbx, unsigned nby) {
    const unsigned xcd = h & 7u, s = h >> 3;
    PreTile t;
    t.rb = (s / nbx) * 8u + xcd;
    t.cb = s % nbx;
    t.on = t.rb < nby;
    return t;
}

// y[r][n] = [ReLU](out_scale * sum_k a[r][k] b[n][k] + bias[n]), per-frame BatchNorm sums in f64, optional finalisation by the
// last workgroup -- the contract of linear_fwd_split (linear_split.hip) with both operands as planes.
template <int NP, int FMT, bool STAMP = false>
__global__ __launch_bounds__(512, 2) void rowgemm_fwd_pre(const unsigned short *__restrict__ a, unsigned a_plane_bytes, int lda,
                                                          const unsigned short *__restrict__ b, unsigned b_plane_bytes, int ldb,
                                                          const float *__restrict__ bias, float *__restrict__ y, int ldy,
                                                          double *__restrict__ stats, const float *__restrict__ row_w,
                                                          long long R, int K, int N, int relu, unsigned *__restrict__ done_counter,
                                                          double fin_eps, float *__restrict__ fin_mean_inv, FrameMap fm,
                                                          float out_scale, unsigned nbx, unsigned nby,
                                                          unsigned long long *__restrict__ stamps, unsigned stamp_block) {
    constexpr int CH = NP * 16, NCH = CH / 8, STAGE = CH * 1024;
    // ALL the LDS of the kernel is this one array (a second __shared__ object beside a DMA staging array makes hipcc wait
    // vmcnt(0) in front of every k-step's first ds_read: cdna_hip_programming.md section 5)
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NSTG * STAGE];
    const PreTile tile = pre_tile(blockIdx.x, nbx, nby);
    if (!tile.on) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 31, lh = lane >> 5;
    const int tm = wv >> 2, tn = wv & 3;
    const long long r0 = (long long)tile.rb * PT;
    const int n0 = tile.cb * PT;
    const int U = K / PKS;
    auto xstamp = [&](int slot) __attribute__((always_inline)) {      // DIAGNOSTIC build: (cycle counter, 100 MHz real-time counter)
        if constexpr (STAMP) {
            if (blockIdx.x == stamp_block && lane == 0) {
                stamps[wv * 512 + 480 + 2 * slot] = __builtin_amdgcn_s_memtime();
                stamps[wv * 512 + 481 + 2 * slot] = __builtin_amdgcn_s_memrealtime();
            }
        }
    };
    xstamp(0);

    // DMA sources: group 1's waves (wv >= 4) move the A half of a stage, group 0's the B half; chunk cc = (wv & 3) + 4 i of the
    // half = (piece q = cc / 8, 32-row block cc % 8).  Lane l fills LDS row l / 2, physical slot l % 2 of its block.
    unsigned src[NCH], src_last[NCH];
    const bool is_a = wv >= 4;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int cc = (wv & 3) + 4 * i;
        const int q = cc >> 3, t = cc & 7;
        const int row = t * 32 + (lane >> 1);
        const int j = (lane & 1) ^ ((row >> 4) & 1);
        if (is_a) {
            long long gr = r0 + row;
            gr = gr < R ? gr : R - 1;
            src[i] = (unsigned)q * a_plane_bytes + (unsigned)(gr * lda + j * 8) * 2u;
        } else {
            int gn = n0 + row;
            gn = gn < N ? gn : N - 1;
            src[i] = (unsigned)q * b_plane_bytes + (unsigned)((long long)gn * ldb + j * 8) * 2u;
        }
        src_last[i] = src[i] + (unsigned)(U - 1) * (PKS * 2);
    }
    f32x16 acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    // (a k16-blocked plane layout -- every DMA instruction reading 1 KB of whole 128-byte lines instead of 32-byte row segments --
    // was measured: 0.665 vs 0.678 ms forward, 0.419 vs 0.416 ms weight gradient: the row-major planes stay)
    pre_mainloop<NP, FMT, 0, STAMP>(smem, acc, (const unsigned char *)a, (const unsigned char *)b, src, src_last, PKS * 2, PKS * 2, U,
                                    wv, lane, PKS, stamps, STAMP && blockIdx.x == stamp_block);
    __syncthreads();                           // the staging buffers become the epilogue's scratch
    xstamp(1);

    if (out_scale != 1.f) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[m][n] *= out_scale;
    }
    // ---- epilogue: bias + ReLU, the per-frame BatchNorm sums in f64, then the stores.  Written for few live registers (the main
    // loop holds 128 accumulators + 72 fragment registers of the 256 a wave has: an earlier form spilled 62 dwords here and
    // every reload cost a memory round trip -- 54 k cycles for the statistics of one tile, measured with the stamps): 32-bit
    // row arithmetic, block-uniform fast paths for interior tiles, row weights loaded 16 at a time.
    const int ir0 = (int)r0, iR = (int)R;
    const bool full = ir0 + PT <= iR && n0 + PT <= N;          // block-uniform: no element of the tile is out of range
    const int col0 = n0 + tn * 64 + li;                        // this lane's column for n = 0 (n = 1: + 32)
    float bsv[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) bsv[n] = bias ? bias[min(col0 + n * 32, N - 1)] : 0.f;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[m][n][r] + bsv[n];
                if (relu) v = fmaxf(v, 0.f);
                acc[m][n][r] = v;
            }
    xstamp(2);
    if (stats) {
        // The per-frame BatchNorm sums, exactly as linear_fwd_split forms them (every term w * v and (w * v) * v in f64), WITHOUT
        // converting accumulators in registers (128 of them + the conversions do not fit a wave: the first forms of this epilogue
        // spilled, or summed four terms in f32 first and then differed from the all-f64 sums by 1e-9): the staging ring is free
        // now, so the tile goes through LDS 64 rows at a time -- every wave writes its 32 x 64 piece of row block m, then thread
        // (column, row half) reads the 32 values of its column and half one by one and accumulates them in f64.
        constexpr int TP = PT + 1;                                // pitch of the LDS copy (floats): conflict-free both ways
        float *s_t = (float *)smem;                               // [64 rows][TP]
        float *s_rw = (float *)(smem + 64 * TP * sizeof(float));  // [PT] row weights of the tile
        int *s_last = (int *)(s_rw + PT);
        const int r_last = min(ir0 + PT, iR) - 1;
        const int s_lo = fm.F == 1 ? 0 : fm_seg_of(fm, ir0), s_hi = fm.F == 1 ? 0 : fm_seg_of(fm, r_last);
        if (tid < PT) s_rw[tid] = row_w ? row_w[min(ir0 + tid, iR - 1)] : 1.f;
        const int sc = tid & (PT - 1), sh = tid >> 8;             // this thread's column of the tile and row half (the waves' tm)
        for (int sg = s_lo; sg <= s_hi; ++sg) {
            const int f = fm.F == 1 ? 0 : (int)fm.seg_frame[sg];
            const int lo = fm.F == 1 ? 0 : fm.bound[sg], hi = fm.F == 1 ? iR : fm.bound[sg + 1];
            double t1 = 0.0, t2 = 0.0;
#pragma unroll 1
            for (int m = 0; m < 4; ++m) {
                __syncthreads();                                  // the previous row block has been read
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v;
                        // (m is a loop variable: select the accumulator statically)
                        v = m == 0 ? acc[0][n][r] : (m == 1 ? acc[1][n][r] : (m == 2 ? acc[2][n][r] : acc[3][n][r]));
                        s_t[(tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * TP + tn * 64 + n * 32 + li] = v;
                    }
                __syncthreads();
                const int rb = sh * 128 + m * 32;                 // tile row of this thread's first value
#pragma unroll 4
                for (int r = 0; r < 32; ++r) {
                    const int gr = ir0 + rb + r;
                    if (gr < iR && gr >= lo && gr < hi) {
                        const double w = (double)s_rw[rb + r], v = (double)s_t[(sh * 32 + r) * TP + sc];
                        t1 += w * v;
                        t2 += w * v * v;
                    }
                }
            }
            if (n0 + sc < N) {
                double *fstats = stats + (size_t)f * MVX_REP * 2 * N + (size_t)(tile.rb % MVX_REP) * 2 * N;
                atomicAdd(fstats + n0 + sc, t1);
                atomicAdd(fstats + N + n0 + sc, t2);
            }
        }
        if (done_counter) bn_finalize_by_last_block(done_counter, nbx * nby, stats, N, fm, fin_eps, fin_mean_inv, s_last);
    }
    xstamp(3);
    // the stores go last (nothing of this workgroup waits behind them)
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int rb = ir0 + tm * 128 + m * 32 + 4 * lh;
        float *yp = y + (size_t)rb * ldy + col0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ro = (r & 3) + 8 * (r >> 2);
            if (full || (rb + ro < iR && col0 < N)) yp[(size_t)ro * ldy] = acc[m][0][r];
            if (full || (rb + ro < iR && col0 + 32 < N)) yp[(size_t)ro * ldy + 32] = acc[m][1][r];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    xstamp(4);
}

// Weight gradient on planes: slab[strip][i][j] = sum over the rows of the strip of A[r][i] * B[r][j] for one 256 x 256 block
// (i over the columns of dz, j over the columns of x).  Strips / slabs / fixed-order reduce as in linear.hip.
template <int NP, int FMT>
__global__ __launch_bounds__(512, 2) void rowgemm_wgrad_pre(const unsigned short *__restrict__ a, unsigned a_plane_bytes, int lda,
                                                            const unsigned short *__restrict__ b, unsigned b_plane_bytes, int ldb,
                                                            float *__restrict__ slabs, long long R, int NA, int NB,
                                                            long long rows_per_strip, unsigned nba, unsigned nbb, unsigned strips,
                                                            int order) {
    constexpr int CH = NP * 16, NCH = CH / 8, STAGE = CH * 1024;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NSTG * STAGE];
    // block -> (strip, output block).  order 1: XCD x = h % 8 takes the strips {8 j + x} whole, so every block that reads a
    // strip's rows shares one L2; order 0: blocks of a strip consecutive (spread over the XCDs round-robin)
    const unsigned nblk = nba * nbb;
    unsigned strip, blk;
    if (order == 1) {
        const unsigned xcd = blockIdx.x & 7u, s = blockIdx.x >> 3;
        strip = (s / nblk) * 8u + xcd;
        blk = s % nblk;
    } else {
        strip = blockIdx.x / nblk;
        blk = blockIdx.x % nblk;
    }
    if (strip >= strips) return;
    const unsigned ia = blk / nbb, ib = blk % nbb;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 31, lh = lane >> 5;
    const int tm = wv >> 2, tn = wv & 3;
    const int a0 = ia * PT, b0 = ib * PT;
    const long long rbeg = (long long)strip * rows_per_strip;
    const long long rend = rbeg + rows_per_strip < R ? rbeg + rows_per_strip : R;
    const int U = (int)((rend - rbeg + PKS - 1) / PKS);
    const int tail_rows = (int)(rend - rbeg) - (U - 1) * PKS;

    // DMA chunk cc = (wv & 3) + 4 i of this wave's half (group 1: A, group 0: B): the 16 rows of one (piece q = cc / 8, 32-column
    // block cc % 8); lane l -> row l / 4, slot l % 4.  Last stage: rows clamped to R - 1 (in-bounds reads; the main loop zeroes
    // rows >= tail_rows in LDS).
    unsigned src[NCH], src_last[NCH];
    const bool is_a = wv >= 4;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int cc = (wv & 3) + 4 * i;
        const int q = cc >> 3, cb = cc & 7;
        int col = (is_a ? a0 : b0) + cb * 32 + (lane & 3) * 8;
        const int lim = (is_a ? NA : NB) - 8;
        col = col < lim ? col : lim;
        const unsigned pb = (unsigned)q * (is_a ? a_plane_bytes : b_plane_bytes);
        const int ld = is_a ? lda : ldb;
        const long long gr = rbeg + (lane >> 2);
        long long gl = rbeg + (long long)(U - 1) * PKS + (lane >> 2);
        gl = gl < R ? gl : R - 1;
        src[i] = pb + (unsigned)(gr * ld + col) * 2u;
        src_last[i] = pb + (unsigned)(gl * ld + col) * 2u;
    }
    f32x16 acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    pre_mainloop<NP, FMT, 1, false>(smem, acc, (const unsigned char *)a, (const unsigned char *)b, src, src_last,
                                    (unsigned)lda * PKS * 2, (unsigned)ldb * PKS * 2, U, wv, lane, tail_rows, nullptr, false);

    float *o = slabs + (size_t)strip * NA * NB;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ai = a0 + tm * 128 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int bi = b0 + tn * 64 + n * 32 + li;
                if (ai < NA && bi < NB) o[(size_t)ai * NB + bi] = acc[m][n][r];
            }
}

// dw[n][k] (+)= scale * sum over the strips of slab[n][k], in strip order (deterministic)
__global__ __launch_bounds__(256) void pre_slab_reduce(const float *__restrict__ slabs, float *__restrict__ dw, size_t total, int strips,
                                                       int accumulate, float scale) {
    for (size_t e = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; e < total; e += (size_t)gridDim.x * 1024) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < strips; ++t) s += *(const f32x4 *)(slabs + (size_t)t * total + e);
        s *= scale;
        if (accumulate) s += *(const f32x4 *)(dw + e);
        *(f32x4 *)(dw + e) = s;
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// C ABI (include/mvx_hip.h)
// ---------------------------------------------------------------------------------------------------------------------
static inline int pre_pieces(int flags) { return (flags & MVX_FLAG_SPLIT_F16) ? 2 : (flags & MVX_FLAG_SPLIT3) ? 3 : 0; }

extern "C" size_t mvx_split_planes_bytes(int64_t rows, int32_t k, int32_t flags) {
    const int np = pre_pieces(flags);
    if (rows <= 0 || k <= 0 || np == 0) return 0;
    return (size_t)np * (size_t)rows * (size_t)k * sizeof(unsigned short);
}

extern "C" int mvx_split_rows(const float *x, int32_t ldx, int64_t rows, int32_t k, void *planes, int32_t flags, float scale,
                              void *stream) {
    const int np = pre_pieces(flags);
    MVX_CHECK_ARG(x && planes && rows >= 0 && k > 0 && k % 4 == 0 && ldx >= k && ldx % 4 == 0 && np != 0);
    MVX_CHECK_ARG((((uintptr_t)x) & 15) == 0 && (((uintptr_t)planes) & 15) == 0);
    if (rows == 0) return MVX_OK;
    hipStream_t st = (hipStream_t)stream;
    const long long total = rows * (k / 4);
    const unsigned grid = (unsigned)(total / 256 + 1 > 8192 ? 8192 : total / 256 + 1);
    if (np == 3)
        hipLaunchKernelGGL((split_rows_kernel<3, 0>), dim3(grid), dim3(256), 0, st, x, ldx, (long long)rows, k,
                           (unsigned short *)planes, (long long)rows * k, k, 1.f);
    else
        hipLaunchKernelGGL((split_rows_kernel<2, 1>), dim3(grid), dim3(256), 0, st, x, ldx, (long long)rows, k,
                           (unsigned short *)planes, (long long)rows * k, k, scale);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

// forward / input-gradient row GEMM on planes.  a: [NP][rows][k] pieces of x (plane stride rows * k), b: [NP][n][k] pieces of
// the weight (for an input gradient: of the transposed weight).  out_scale multiplies the accumulators (fp16 pieces: the
// inverse of the operands' power-of-two scales; 1 otherwise).
extern "C" int mvx_linear_forward_pre_frames(const void *a_planes, const void *b_planes, const float *bias, float *y, int32_t ldy,
                                             double *stats, const float *row_w, int64_t rows, int32_t k, int32_t n, int32_t flags,
                                             float out_scale, uint32_t *fin_counter, double fin_eps, float *fin_mean_inv,
                                             const mvx_frames_t *frames_host, int32_t row_kind, void *stream) {
    const int np = pre_pieces(flags);
    MVX_CHECK_ARG(a_planes && b_planes && y && rows >= 0 && k > 0 && n > 0 && ldy >= n && np != 0);
    MVX_CHECK_ARG(k % PKS == 0 && (((uintptr_t)a_planes) & 15) == 0 && (((uintptr_t)b_planes) & 15) == 0);
    MVX_CHECK_ARG((size_t)np * rows * k * 2 < (1ull << 32) && (size_t)np * n * k * 2 < (1ull << 32));
    hipStream_t st = (hipStream_t)stream;
    FrameMap fm;
    MVX_CHECK_ARG(mvx_build_frame_map(fm, frames_host, row_kind, rows, (double)rows));
    if (stats && !(flags & MVX_FLAG_PREZEROED)) {
        hipError_t e = hipMemsetAsync(stats, 0, sizeof(double) * MVX_REP * 2 * n * fm.F, st);
        if (e != hipSuccess) return (int)e;
    }
    if (rows == 0) return MVX_OK;
    const unsigned nbx = mvx_cdiv(n, PT), nby = mvx_cdiv(rows, PT);
    const dim3 grid(8u * ((nby + 7u) / 8u) * nbx);
    const unsigned aps = (unsigned)((size_t)rows * k * 2), bps = (unsigned)((size_t)n * k * 2);
    const int relu = flags & MVX_FLAG_RELU;
    if (np == 3)
        hipLaunchKernelGGL((rowgemm_fwd_pre<3, 0>), grid, dim3(512), 0, st, (const unsigned short *)a_planes, aps, k,
                           (const unsigned short *)b_planes, bps, k, bias, y, ldy, stats, row_w, (long long)rows, k, n, relu,
                           fin_counter, fin_eps, fin_mean_inv, fm, out_scale, nbx, nby, (unsigned long long *)nullptr, 0u);
    else
        hipLaunchKernelGGL((rowgemm_fwd_pre<2, 1>), grid, dim3(512), 0, st, (const unsigned short *)a_planes, aps, k,
                           (const unsigned short *)b_planes, bps, k, bias, y, ldy, stats, row_w, (long long)rows, k, n, relu,
                           fin_counter, fin_eps, fin_mean_inv, fm, out_scale, nbx, nby, (unsigned long long *)nullptr, 0u);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

// strips of the weight gradient: one workgroup per CU and (about) one round of workgroups
static inline void pre_wgrad_shape(int64_t rows, int32_t k, int32_t n, long long &strips, long long &per) {
    const long long blocks = (long long)mvx_cdiv(n, PT) * mvx_cdiv(k, PT);
    strips = 256 / blocks;
    if (strips < 1) strips = 1;
    per = ((rows + strips - 1) / strips + PKS - 1) / PKS * PKS;
    if (per < PKS) per = PKS;
    strips = (rows + per - 1) / per;
    if (strips < 1) strips = 1;
}

extern "C" size_t mvx_linear_wgrad_pre_workspace_bytes(int64_t rows, int32_t k, int32_t n) {
    if (rows <= 0 || k <= 0 || n <= 0) return 0;
    long long strips, per;
    pre_wgrad_shape(rows, k, n, strips, per);
    return (size_t)strips * n * k * sizeof(float);
}

static int wgrad_pre_launch(const unsigned char *x_planes, const unsigned char *dz_planes, float *dw, int64_t plane_rows,
                            int64_t row_lo, int64_t rows, int32_t k, int32_t n, int32_t flags, float out_scale, void *workspace,
                            size_t workspace_bytes, void *stream) {
    const int np = pre_pieces(flags);
    MVX_CHECK_ARG(x_planes && dz_planes && dw && workspace && rows > 0 && k > 0 && n > 0 && np != 0);
    MVX_CHECK_ARG(row_lo >= 0 && row_lo + rows <= plane_rows);
    MVX_CHECK_ARG(k % 32 == 0 && n % 32 == 0 && (((uintptr_t)dw) & 15) == 0 && ((size_t)n * k) % 4 == 0);
    MVX_CHECK_ARG((size_t)np * plane_rows * k * 2 < (1ull << 32) && (size_t)np * plane_rows * n * 2 < (1ull << 32));
    long long strips, per;
    pre_wgrad_shape(rows, k, n, strips, per);
    MVX_CHECK_ARG(workspace_bytes >= (size_t)strips * n * k * sizeof(float));
    hipStream_t st = (hipStream_t)stream;
    const unsigned nba = mvx_cdiv(n, PT), nbb = mvx_cdiv(k, PT);
    // planes are plane_rows apart; the rows of this call start row_lo into every plane (k, n multiples of 32: 16-byte aligned)
    const unsigned aps = (unsigned)((size_t)plane_rows * n * 2), bps = (unsigned)((size_t)plane_rows * k * 2);
    const unsigned short *a = (const unsigned short *)(dz_planes + (size_t)row_lo * n * 2);
    const unsigned short *b = (const unsigned short *)(x_planes + (size_t)row_lo * k * 2);
    const int order = (flags & MVX_FLAG_PRE_XCD_STRIPS) ? 1 : 0;
    const unsigned nwg = order ? 8u * (unsigned)((strips + 7) / 8) * nba * nbb : (unsigned)strips * nba * nbb;
    if (np == 3)
        hipLaunchKernelGGL((rowgemm_wgrad_pre<3, 0>), dim3(nwg), dim3(512), 0, st, a, aps, n, b, bps, k, (float *)workspace,
                           (long long)rows, n, k, per, nba, nbb, (unsigned)strips, order);
    else
        hipLaunchKernelGGL((rowgemm_wgrad_pre<2, 1>), dim3(nwg), dim3(512), 0, st, a, aps, n, b, bps, k, (float *)workspace,
                           (long long)rows, n, k, per, nba, nbb, (unsigned)strips, order);
    MVX_LAUNCH_CHECK();
    const size_t total = (size_t)n * k;
    hipLaunchKernelGGL(pre_slab_reduce, dim3((unsigned)(total / 1024 + 1 > 2048 ? 2048 : total / 1024 + 1)), dim3(256), 0, st,
                       (const float *)workspace, dw, total, (int)strips, (flags & MVX_FLAG_ACCUMULATE) ? 1 : 0, out_scale);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" int mvx_linear_wgrad_pre(const void *x_planes, const void *dz_planes, float *dw, int64_t rows, int32_t k, int32_t n,
                                    int32_t flags, float out_scale, void *workspace, size_t workspace_bytes, void *stream) {
    return wgrad_pre_launch((const unsigned char *)x_planes, (const unsigned char *)dz_planes, dw, rows, 0, rows, k, n, flags,
                            out_scale, workspace, workspace_bytes, stream);
}

// The same over the rows [row_lo, row_hi) of planes that hold plane_rows rows each (the workspace of the whole range suffices):
// the weight gradient of a row range whose dz planes are already written (mvx_bn_relu_backward_planes_part_frames).
extern "C" int mvx_linear_wgrad_pre_rows(const void *x_planes, const void *dz_planes, float *dw, int64_t plane_rows, int64_t row_lo,
                                         int64_t row_hi, int32_t k, int32_t n, int32_t flags, float out_scale, void *workspace,
                                         size_t workspace_bytes, void *stream) {
    MVX_CHECK_ARG(row_hi > row_lo);
    return wgrad_pre_launch((const unsigned char *)x_planes, (const unsigned char *)dz_planes, dw, plane_rows, row_lo,
                            row_hi - row_lo, k, n, flags, out_scale, workspace, workspace_bytes, stream);
}

// DIAGNOSTIC (tools/stamps_rows_pre.py): the bf16x6 forward with s_memtime stamps of one workgroup; not part of the C ABI header
extern "C" int mvx_debug_rowgemm_fwd_pre_stamps(const void *a_planes, const void *b_planes, float *y, int64_t rows, int32_t k, int32_t n,
                                                unsigned long long *stamps, uint32_t stamp_block, void *stream, double *stats) {
    FrameMap fm;
    if (!mvx_build_frame_map(fm, nullptr, MVX_ROWS_SINGLE, rows, (double)rows)) return MVX_EINVAL;
    const unsigned nbx = mvx_cdiv(n, PT), nby = mvx_cdiv(rows, PT);
    const dim3 grid(8u * ((nby + 7u) / 8u) * nbx);
    hipLaunchKernelGGL((rowgemm_fwd_pre<3, 0, true>), grid, dim3(512), 0, (hipStream_t)stream, (const unsigned short *)a_planes,
                       (unsigned)((size_t)rows * k * 2), k, (const unsigned short *)b_planes, (unsigned)((size_t)n * k * 2), k,
                       (const float *)nullptr, y, n, stats, (const float *)nullptr, (long long)rows, k, n, 1,
                       (unsigned *)nullptr, 0.0, (float *)nullptr, fm, 1.f, nbx, nby, stamps, stamp_block);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}
