/* libmvx_hip -- C ABI of the MI355X-native MVXNet hot path (gfx950).
 *
 * This library stands where the reference's pybind11 extension stood
 * (modules/Extension.py:1-3 -> cpp/voxelutil.cpp:362-368, entry `_group`) and additionally
 * carries the GPU kernels that the reference obtained implicitly from ATen/cuDNN for the
 * modules on the hot path.
 *
 * Conventions (every entry point):
 *   - plain C types only; every data pointer is a DEVICE pointer unless its name ends in
 *     `_host`; `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - returns 0 on success, a negative MVX_E* code for an argument error, or a positive
 *     hipError_t from the launch;
 *   - stream-ordered, no hidden synchronisation, no persistent allocation: scratch memory
 *     comes from the caller (`*_workspace_bytes` queries), so calls can be graph-captured;
 *   - re-entrant and stream-ordered: no entry point keeps state between calls.  Process-wide exceptions, none of them in a
 *     data path: the launch-shape tuning values of mvx_tuning_set (read at launch time; the tests set them to force a
 *     kernel shape) and the diagnostic counter behind mvx_launch_count.  One per-THREAD exception: the operand ranges bound
 *     by mvx_split_operand_amax apply to the calling thread's next split-arithmetic launch and are cleared by it.
 */
#ifndef MVX_HIP_H
#define MVX_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Per-channel BatchNorm accumulators are kept in MVX_STATS_REPLICAS replicas (workgroup b adds to
 * replica b mod R) so that the f64 atomics of concurrently finishing workgroups do not pile up on
 * one address; every `stats` buffer below is f64 [MVX_STATS_REPLICAS][2][channels] and
 * mvx_bn_finalize sums the replicas. */
#define MVX_STATS_REPLICAS 32

/* Bits of the `flags` arguments.  (Entry points that used to take `relu` take `flags`; bit 0 keeps
 * the old meaning.) */
#define MVX_FLAG_RELU 1        /* apply ReLU in the epilogue */
#define MVX_FLAG_PREZEROED 2   /* the stats / scratch accumulators passed in are already zero: skip the memset
                                  (lets a caller clear all accumulators of a frame with ONE fill) */
#define MVX_FLAG_ACCUMULATE 4  /* add the gradient to the destination instead of overwriting it */
#define MVX_FLAG_CONV2D 8      /* mvx_conv3d_wgrad: dw is a 2-D kernel gradient [cout][cin][3][3] (din = dout = 1, pad_d = 1) */
#define MVX_FLAG_TAPS2 16      /* mvx_conv2d_*: only the 2x2 tap window {0,1}^2 carries weight (stride-2 conv on the space-to-depth image) */
#define MVX_FLAG_BG_TAPS 32    /* mvx_conv3d_forward_bg_frames: bg_pre is the buffer of mvx_conv3d_background_taps_frames ([planes][cout] totals |
                                  [planes][3][cout] per-depth-tap constants | [planes][9][cout] image-border position classes): in interior
                                  tiles a depth tap whose source halo holds no active site is not executed (its constant is added in the
                                  epilogue), and a border tile without any active source is filled from the class constants (exact rewrites) */
#define MVX_FLAG_SPLIT 64      /* mvx_linear_forward*, mvx_linear_wgrad, mvx_conv3d_wgrad, mvx_conv3d_wgrad_bg*, mvx_conv2d_wgrad_frames (conv3d_wgrad4s: cin a
                                  multiple of 64): bf16x3 split arithmetic (three bf16 MFMAs per product, f32 accumulate, ~2e-5 per
                                  product: the row-GEMM side of `convmath: bf16x3`) for layers with n > 64, k % 4 == 0, 16-byte aligned
                                  operands and a row-major weight (w_transposed = 0); other calls run the exact-f32 kernel */

#define MVX_FLAG_SPLIT3 128    /* with MVX_FLAG_SPLIT, and on the *_split entry points: THREE bf16 pieces per f32 operand and six bf16 MFMAs per product
                                  ("bf16x6": hi + mid + lo is the operand exactly, dropped cross terms < 2^-25: fp32-grade accuracy) instead of two / three */

#define MVX_FLAG_PRE_XCD_STRIPS 4096  /* mvx_linear_wgrad_pre: every block of a row strip on the same XCD (shared L2) instead of consecutive block ids */
#define MVX_FLAG_SPLIT_F16 512  /* with MVX_FLAG_SPLIT, and on the *_split entry points: TWO fp16 pieces per f32 operand (22 mantissa bits) and three
                                  fp16 MFMAs per product ("fp16x3": fp32-grade accuracy at the matrix work of bf16x3) -- for operands inside fp16's
                                  range: values above 65,504 overflow and below ~1e-4 lose relative precision, so callers scale by powers of two */

#define MVX_FLAG_AMAX_COARSE 1024 /* mvx_linear_forward*: the x operand bound by mvx_split_operand_amax is a FORWARD input from outside the library
                                  (sampled image features): its scale exponent is rounded down to a multiple of 8 binades, so that a frame set and
                                  one of its frames -- whose maxima differ -- scale, and therefore round, alike */

#define MVX_FLAG_NO_BG_FILL 2048  /* mvx_sparse_conv_output_frames: tiles (8 x 16 sites) whose 3 x 3 tile neighbourhood holds no voxel in any source
                                  plane are NOT written -- every site of them is ReLU(bias), which the tile-restricted consumers
                                  (mvx_bn_apply_tiles_frames, mvx_bn_relu_backward_tiles_frames) never read; the BatchNorm sums count them
                                  in closed form either way */

#define MVX_OK 0
#define MVX_EINVAL (-1)   /* bad argument (null pointer, size, unsupported combination) */
#define MVX_ESIZE (-2)    /* a size exceeds what the kernel supports */

/* ------------------------------------------------------------------------------------------
 * Frame sets.  The reference is batch-1 (config.yml:18, modules/voxelnet/VoxelNet.py:19): a batch of B frames is B
 * independent forwards with PER-FRAME BatchNorm statistics and summed parameter gradients.  The
 * `*_frames` entry points run all frames of a step through ONE launch per layer: row matrices hold the frames back to
 * back, grids stack them along the depth axis ([n_frames * planes][h][w][c]), and every per-frame quantity (statistics,
 * mean / inverse std, background constants, counters) gets a leading frame dimension.  Parameter gradients are summed
 * over the frames by construction.  The descriptor lives on the HOST and is copied by value into the launches.
 *
 *   real_off[f] .. real_off[f+1]   compact REAL rows of frame f (points that survived the T cap), all frames back to back
 *   vox_off[f]  .. vox_off[f+1]    voxels of frame f
 *   t                              samples per voxel: frame f stands for (vox_off[f+1]-vox_off[f]) * t dense rows
 * Row layouts (`row_kind`):
 *   MVX_ROWS_SINGLE  one frame, no descriptor (what the entry points without `_frames` use)
 *   MVX_ROWS_FUSION  [real rows of all frames][one shared padded row per frame]        (fusion MLP, imhead/Pipe.py:84-104)
 *   MVX_ROWS_VFE     [real rows of all frames][one padded row per voxel, all frames]   (VFE stack, voxelnet/Pipe.py:5-29)
 *   MVX_ROWS_VOXELS  [one row per voxel, all frames]
 *   MVX_ROWS_GRID    [n_frames][rows / n_frames] equal shares (channels-last grids)
 *   MVX_ROWS_REAL    [real rows of all frames] (the sampler's output before the shared padded rows are appended)
 */
#define MVX_MAX_FRAMES 16
#define MVX_ROWS_SINGLE 0
#define MVX_ROWS_FUSION 1
#define MVX_ROWS_VFE 2
#define MVX_ROWS_VOXELS 3
#define MVX_ROWS_GRID 4
#define MVX_ROWS_REAL 5        /* [real rows of all frames] only */
typedef struct {
    int32_t n_frames;
    int32_t t;
    int32_t real_off[MVX_MAX_FRAMES + 1];
    int32_t vox_off[MVX_MAX_FRAMES + 1];
} mvx_frames_t;

/* ABI version; bumped whenever a signature changes. */
int mvx_abi_version(void);
/* tuning values of the library (process-wide, not stream-ordered; meant for benchmarks and tests).  Keys:
 *   MVX_TUNE_SPLIT16_MIN_UNITS  the bf16x3 gather uses 16 x 16-site workgroup units when a launch has at least this many of them
 *                               (default 768), else 8 x 16-site units; 0 = always 16 x 16, a huge value = never
 *   MVX_TUNE_GATHER_NARROW_MAX_UNITS  the f32 gather (conv3d / conv2d forward and dgrad) runs a launch with fewer 64-channel
 *                               workgroup units than this as twice as many 32-channel units; 0 = never, negative (default) = fewer than
 *                               1.5 units per CU, i.e. while the doubled units are all resident at once
 *   MVX_TUNE_ROWGEMM_K128       1 (default): split-arithmetic row layers with k = 128 and n a multiple of 128 run the
 *                               weights-resident streaming kernel (csrc/rowgemm_k128.hip; same numbers); 0: linear_fwd_split */
#define MVX_TUNE_SPLIT16_MIN_UNITS 1
#define MVX_TUNE_GATHER_NARROW_MAX_UNITS 2
#define MVX_TUNE_ROWGEMM_K128 3
int mvx_tuning_set(int32_t key, int64_t value);

/* fp16x3 arithmetic (MVX_FLAG_SPLIT_F16): operand ranges.  An fp16 piece pair covers an f32 operand to 2^-22 while the low piece
 * is a normal fp16 number, i.e. for magnitudes in [2^-3, 65504]; activations behind a BatchNorm and weights (scaled by a fixed
 * 2^8 inside the library) sit there, gradients (1e-3 ... 1e-8) and foreign feature maps need not.  For those the caller binds
 * the device address of ONE float = max |value| of the tensor, and the kernel multiplies the operand by the power of two that
 * brings that maximum into [2^14, 2^15) and the f32 accumulator by its inverse -- powers of two change no rounding, so the
 * result is the same function of the inputs.  amax_a / amax_b belong to the first / second f32 operand of the calling
 * thread's NEXT launch that takes MVX_FLAG_SPLIT_F16 (convolution forward / input gradient and row GEMM: a = in / dz / x;
 * weight gradients: a = in / x, b = dz); NULL = that operand is not scaled.  The launch consumes the binding.  The floats are
 * read on the device in stream order: they may be written by an earlier kernel of the same stream.  Sources:
 * mvx_bn_relu_backward_frames and mvx_bn_relu_backward_tiles_frames
 * (dz_amax), mvx_tensor_amax for any other tensor.  Bound ranges are ignored by the other arithmetics. */
int mvx_split_operand_amax(const float *amax_a, const float *amax_b);
/* amax[0] = max(amax[0], max |x[i]|), i < n; amax is zeroed first unless MVX_FLAG_PREZEROED.  x 16-byte aligned. */
int mvx_tensor_amax(const float *x, int64_t n, float *amax, int32_t flags, void *stream);
/* Range guard of the fp16-piece arithmetic: weights are cut times 2^8, so |w| must stay below 65504 / 256 = 255.9.
 * status[0] |= MVX_STATUS_F16_WEIGHT_RANGE when an element of w (f32 [n]) does not (or is not finite); status is a device
 * word the caller checks with its other data-dependent status words (once per step: no host read inside the step). */
#define MVX_STATUS_F16_WEIGHT_RANGE 8
int mvx_split_f16_weight_check(const float *w, int64_t n, int32_t *status, void *stream);
/* Diagnostics: number of kernel launches the library has issued since it was loaded (fills excluded). */
uint64_t mvx_launch_count(void);

/* ------------------------------------------------------------------------------------------
 * Voxelizer.  Replaces cpp/voxelutil.cpp:325-360 (`_group`) together with the Python around
 * it: modules/data/Preprocessing.py:75-116 (`group`, out_channels = 9) and :57-73 (`group_`,
 * out_channels = 7).
 *
 * Frames are batched with a fixed stride: frame f owns points  pcd[f*cap_points ...] and
 * outputs [f*cap_voxels ...]; the live point count of each frame is read from DEVICE memory
 * (n_points[f]) so a preceding GPU crop can feed it without a host round trip.
 *
 *   pcd        f32 [F][cap_points][ncol]   ncol >= 4: x y z r (row col)
 *   perm       i32 [F][cap_points] or NULL shuffle permutation (Preprocessing.py:86 is an
 *                                          input here: stream position s reads point perm[s])
 *   n_points   i32 [F]
 *   ext_idx    i32 [F][cap_points][3] or NULL: precomputed voxel indices in STREAM order, the
 *                                          `idx` argument of the reference `_group`
 *                                          (voxelutil.cpp:325); NULL = computed here
 *   lo[3], size[3]                         range minimum and voxel size, f64 (Config.py:7)
 *   T                                      samplesPerVoxel (<= 64)
 *   out_channels                           9: x y z dx dy dz r row col, centroid = sequential
 *                                             f64 sum / count (Preprocessing.py:112-115)
 *                                          7: x y z dx dy dz r, centroid = f32 sum / count in
 *                                             f64 (Preprocessing.py:71-72)
 *   voxels     f32 [F][cap_voxels][T][out_channels]   (the reference's f64 rounded once to
 *                                          f32, which is what train.py:125 feeds the model)
 *   coords     i64 [F][cap_voxels][4]      (0, ix, iy, iz)  == train.py:119 layout
 *   counts     i32 [F][cap_voxels]         kept points per voxel (<= T)
 *   n_voxels   i32 [F]
 *   status     i32 [1]                     OR-ed flags: bit0 = an index fell outside the
 *                                          21-bit key range, bit1 = cap_voxels exceeded
 * cap_voxels >= cap_points always suffices.
 *
 * mvx_voxelize_frames: the same with the batch layout as an option.  concat = 0: as above.  concat != 0: the voxels of
 *   all frames are written BACK TO BACK in frame order -- voxels [cap_voxels][T][C], coords [cap_voxels][4] with
 *   coords[:,0] = frame index (the batch column of train.py:119), counts [cap_voxels]; cap_voxels is then the TOTAL
 *   capacity (n_frames * cap_points always suffices).  vox_off (optional) i32 [n_frames + 1] on the device = voxel
 *   offsets of the frames (vox_off[n_frames] = total).  n_frames <= 255.
 * Three launches (insert, one look-back scan over all frames, gather) and two memsets per call.
 */
size_t mvx_voxelize_workspace_bytes(int32_t n_frames, int32_t cap_points);

int mvx_voxelize(const float *pcd, const int32_t *perm, const int32_t *n_points,
                 const int32_t *ext_idx, int32_t n_frames, int32_t cap_points, int32_t ncol,
                 double lo_x, double lo_y, double lo_z,
                 double size_x, double size_y, double size_z,
                 int32_t T, int32_t out_channels, int32_t cap_voxels,
                 float *voxels, int64_t *coords, int32_t *counts, int32_t *n_voxels,
                 int32_t *status, void *workspace, size_t workspace_bytes, void *stream);
int mvx_voxelize_frames(const float *pcd, const int32_t *perm, const int32_t *n_points,
                        const int32_t *ext_idx, int32_t n_frames, int32_t cap_points, int32_t ncol,
                        double lo_x, double lo_y, double lo_z,
                        double size_x, double size_y, double size_z,
                        int32_t T, int32_t out_channels, int32_t cap_voxels, int32_t concat,
                        float *voxels, int64_t *coords, int32_t *counts, int32_t *n_voxels, int32_t *vox_off,
                        int32_t *status, void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * Sparse voxel rows <-> dense grid.  Replaces VoxelNet.reindex (modules/voxelnet/VoxelNet.py:16-22,
 * `res[b,:,iz,ix,iy] = x[v,:]`) and its autograd gather.  The grid is channels-last:
 *   grid f32 [d=iz][h=ix][w=iy][channels]   (logical NCDHW (1,C,D,H,W) with C innermost)
 *   feat f32 [n_voxels][channels], coords i64 [n_voxels][4] = (b, ix, iy, iz)
 * zero_grid != 0 clears the whole grid first (the reference allocates zeros every call).
 * status (optional) bit0 = a coordinate fell outside the grid (row skipped).
 * occupancy (optional) i32 [d][ceil(h/tile_h)][ceil(w/tile_w)]: number of voxels per tile, for the
 * input-sparse first convolution (tile shape from mvx_conv3d_tile_shape).
 * site_bits (optional) u32 [d][h][ceil(w/32)]: one bit per site that holds a voxel.
 */
int mvx_scatter_voxels(const float *feat, const int64_t *coords, float *grid, int32_t n_voxels,
                       int32_t channels, int32_t d, int32_t h, int32_t w, int32_t zero_grid,
                       int32_t *status, int32_t *occupancy, int32_t tile_h, int32_t tile_w,
                       uint32_t *site_bits, void *stream);
int mvx_gather_voxels(const float *grid, const int64_t *coords, float *feat, int32_t n_voxels,
                      int32_t channels, int32_t d, int32_t h, int32_t w, void *stream);

/* CML output -> bird's-eye-view map: the reshape of VoxelNet.py:36, (1,C,D,H,W) -> (1,C*D,H,W).
 *   cl  f32 [d][h][w][channels] (channels-last),  bev f32 [channels*d][h][w], bev channel = c*d_total + d
 *   reverse = 0: cl -> bev (forward);  reverse = 1: bev -> cl (its gradient).  `cl` is written then. */
int mvx_cl_to_bev(const float *cl, float *bev, int32_t d, int32_t h, int32_t w, int32_t channels,
                  int32_t reverse, void *stream);

/* ------------------------------------------------------------------------------------------
 * BatchNorm with batch statistics, fused with the preceding ReLU.  Replaces the
 * nn.BatchNorm2d/3d(affine=False, track_running_stats=False) calls of
 * modules/layers/Blocks.py:10,16,25,29 (forward) and their autograd (backward).
 * All matrices are row-major [rows][channels] f32 (channels-last), channels % 4 == 0.
 *
 *   mvx_row_stats       stats (replicated, see MVX_STATS_REPLICAS) = per-channel (sum, sum of squares)
 *   mvx_bn_finalize     mean_inv f32 [2][C] = (mean, 1/sqrt(biased var + eps)), count = #rows
 *   mvx_bn_apply        out = (y - mean) * inv          (out may alias y)
 *   mvx_bn_relu_backward
 *        given dyhat = dL/d(BN output) and y = ReLU output (BN input):
 *        dz = (y > 0) ? inv * (dyhat - mean(dyhat) - yhat * mean(dyhat * yhat)) : 0
 *        dbias (optional) f32 [C] = column sums of dz;  scratch: mvx_bn_backward_scratch_bytes(C)
 *        bytes (replicated f64 accumulators);  dz may alias dyhat.
 *        row_w (optional) f32 [rows]: row r stands for row_w[r] identical rows of the dense tensor
 *        and dyhat[r] is already the SUM of their gradients; `count` is the dense row count.
 */
int mvx_row_stats(const float *y, double *stats, int64_t rows, int32_t channels, void *stream);
int mvx_bn_finalize(const double *stats, double count, double eps, float *mean_inv, int32_t channels,
                    void *stream);
int mvx_bn_apply(const float *y, const float *mean_inv, float *out, int64_t rows, int32_t channels,
                 void *stream);
size_t mvx_bn_backward_scratch_bytes(int32_t channels);
int mvx_bn_relu_backward(const float *dyhat, const float *y, const float *mean_inv, double count,
                         float *dz, float *dbias, double *scratch, const float *row_w, int64_t rows,
                         int32_t channels, int32_t flags, void *stream);

/* ------------------------------------------------------------------------------------------
 * Dense 3x3x3 convolution on the matrix cores (fp32 MFMA), one frame, channels-last
 * activations [D][H][W][C].  Replaces the nn.Conv3d forward/backward that CML obtains from
 * ATen/cuDNN (modules/voxelnet/Pipe.py:36-43, modules/layers/Blocks.py:24,28): kernel 3,
 * stride (stride_d,1,1), padding (pad_d,1,1); cin % 32 == 0, cout % 64 == 0.
 *
 *   mvx_conv3d_pack_weights  torch layout W[cout][cin][3][3][3] -> kernel layout
 *                            (for_dgrad bit 0 = 0: forward operand, 1: transposed/flipped operand; bit 1 set: the source
 *                            is a 2-D kernel W[cout][cin][3][3] = the middle depth slice -- with din = dout = 1 and
 *                            pad_d = 1 the same kernels then evaluate nn.Conv2d 3x3, stride 1, padding 1: the RPN
 *                            blocks of modules/voxelnet/Pipe.py:45-75, next scope row)
 *   mvx_conv3d_forward       out = [ReLU](conv(in) + bias); stats (optional, replicated f64 [R][2][cout]) =
 *                            per-channel (sum, sum of squares) of `out` for the BatchNorm that
 *                            follows (Blocks.py:28-29)
 *   work_counter (optional, forward / dgrad and their _bg / _tiles forms): u32 [1] holding ZERO; the launch then uses
 *                            the persistent form of the kernel -- two workgroups per CU pull (tile, plane, channel block)
 *                            units from this counter until none is left: no tail round of idle CUs, constant-fill tiles do
 *                            not unbalance the workgroups.  NULL: one workgroup per unit.
 *   mvx_conv3d_dgrad         dx [din][h][w][cin] from dz [dout][h][w][cout]
 *   mvx_conv3d_wgrad         dw in torch layout [cout][cin][3][3][3] ([cout][cin][3][3] with MVX_FLAG_CONV2D);
 *                            cout % 64 == 0 when cin % 64 == 0, else cout == 64
 */
size_t mvx_conv3d_packed_weight_bytes(int32_t cout, int32_t cin);
int mvx_conv3d_pack_weights(const float *w, float *wpk, int32_t cout, int32_t cin, int32_t for_dgrad,
                            void *stream);
void mvx_conv3d_tile_shape(int32_t *tile_h, int32_t *tile_w);
int mvx_conv3d_forward(const float *in, const float *wpk, const float *bias, float *out, double *stats,
                       int32_t din, int32_t dout, int32_t h, int32_t w, int32_t cin, int32_t cout,
                       int32_t stride_d, int32_t pad_d, int32_t flags, uint32_t *work_counter, void *stream);
int mvx_conv3d_dgrad(const float *dz, const float *wpk_dgrad, float *dx, int32_t din, int32_t dout,
                     int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t stride_d, int32_t pad_d,
                     uint32_t *work_counter, void *stream);
size_t mvx_conv3d_wgrad_workspace_bytes(int32_t h, int32_t w, int32_t cin, int32_t cout);
int mvx_conv3d_wgrad(const float *in, const float *dz, float *dw, int32_t din, int32_t dout, int32_t h,
                     int32_t w, int32_t cin, int32_t cout, int32_t stride_d, int32_t pad_d, int32_t flags,
                     void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * Background rewrite of the CML stack (csrc/activity.hip).  The grid VoxelNet.reindex fills
 * (modules/voxelnet/VoxelNet.py:16-22) is zero outside the voxel sites and every CML block is
 * Conv3d -> ReLU -> BatchNorm without affine (modules/voxelnet/Pipe.py:31-43, modules/layers/Blocks.py:20-29),
 * so a site without a voxel in its receptive field holds one per-(plane, channel) constant after each
 * layer.  These entry points describe that background and let forward and wgrad skip it -- exact
 * rewrites of the dense arithmetic, no approximation:
 *
 *   mvx_activity_dilate   dst_mask u8 [dout][h][w] = 1 where the 3x3x3 receptive field holds an active source
 *                         site (src: i32 index grid [din][h][w], -1 = empty, or u8 mask), or, with mark_border,
 *                         where the in-plane window leaves the image (zero padding is not the background);
 *                         dst_halo_flags (optional) i32 [dout][tiles]: the (8+2)x(16+2) halo of the tile holds
 *                         an active site of that plane (tiles as mvx_conv3d_tile_shape)
 *   mvx_conv3d_background bg_pre f32 [dout][cout] = conv of the constant input c_in f32 [din][cin] at an interior
 *                         site (valid depth taps of each plane only); w in torch layout
 *   mvx_bn_background     y_bg = [ReLU](bg_pre + bias), c_out = (y_bg - mean) * inv: the normalised background
 *                         of the layer output, bit-equal to what mvx_bn_apply writes at background sites
 *   mvx_conv3d_forward_bg mvx_conv3d_forward on a source with background: tiles whose halo flags are clear
 *                         (and, with border_active, that do not touch the image border) are filled with the
 *                         constant; background SITES (out_mask == 0) inside computed tiles get it too.
 *                         exec_stages (optional) u64 [1] += executed (depth tap, 32-channel) stages of 4.7 MFLOP
 *   mvx_plane_tap_sums    tap_sums f32 [planes][9][channels]: for each in-plane tap (a,b) the sum of dz over the sites
 *                         whose tap source (y+a-1, x+b-1) lies inside the image (f64 accumulation).  With tile_flags
 *                         + inactive_sums f32 [planes][channels] (both or neither): dz is only defined on the flagged
 *                         tiles (which must include every border tile); the rest of each plane contributes inactive_sums
 *   mvx_tile_dilate_flags out[d][t] = self[d][t] | any 3x3 tile neighbour flagged in an input plane tied to d by a depth
 *                         tap: the tiles of a layer's output gradient that a restricted backward of its consumer reads
 *   mvx_conv3d_wgrad_bg   dw = sum (in - c_in) (x) dz over the tiles with a set halo flag + c_in (x) tap_sums, equal to
 *                         mvx_conv3d_wgrad
 *   mvx_conv3d_input_grad_sums  plane_grad_sums f32 [din][cin] = per input plane, the sum over all its sites of the
 *                         input gradient (what mvx_conv3d_dgrad would write), from tap_sums in closed form
 *   mvx_conv3d_dgrad_tiles      mvx_conv3d_dgrad for the output tiles with dx_tile_flags i32 [din][tiles] set; the
 *                         other tiles of dx are left untouched
 *   mvx_bn_relu_backward_tiles  mvx_bn_relu_backward of a layer whose output is the background (y_bg, c_bg per plane)
 *                         outside the flagged tiles and whose incoming gradient dyhat is only valid ON them:
 *                         the batch sums take the rest from plane_grad_sums; dz is written on the flagged tiles only;
 *                         dz_inactive_sums (optional) f32 [planes][channels] = sum of dz over the other tiles (closed form)
 */
int mvx_activity_dilate(const void *src, int32_t src_is_index, int32_t din, int32_t dout, int32_t h, int32_t w,
                        int32_t stride_d, int32_t pad_d, int32_t mark_border, uint8_t *dst_mask,
                        int32_t *dst_halo_flags, int32_t *dst_tile_flags, void *stream);
/* BatchNorm apply of a layer output with a background: (y - mean) * inv on the tiles whose flag is set (tiles as
 * mvx_conv3d_tile_shape, flags i32 [n_frames * planes][tiles]), the normalised background constant c_bg f32
 * [n_frames * planes][channels] (mvx_bn_background) everywhere else, without reading y there.  Bit-identical to mvx_bn_apply. */
int mvx_bn_apply_tiles_frames(const float *y, const float *mean_inv, const float *c_bg, const int32_t *tile_flags, float *out,
                              int32_t planes, int32_t h, int32_t w, int32_t channels, int32_t n_frames, void *stream);
/* ... that leaves the background tiles no consumer reads unwritten.  read_flags [n_frames * planes][tiles]: mvx_tile_read_flags_frames
 * of the CONSUMING layer (in_halo_flags = this tensor's halo flags, din / dout / stride_d / pad_d = the consumer's depth geometry):
 * non-zero = the consumer's background-aware forward or weight gradient may read the tile (a superset: the 3 x 3 tile neighbourhoods
 * of the output tiles it computes, in every valid source plane). */
int mvx_bn_apply_tiles_read_frames(const float *y, const float *mean_inv, const float *c_bg, const int32_t *tile_flags,
                                   const int32_t *read_flags, float *out, int32_t planes, int32_t h, int32_t w, int32_t channels,
                                   int32_t n_frames, void *stream);
int mvx_tile_read_flags_frames(const int32_t *in_halo_flags, int32_t din, int32_t dout, int32_t h, int32_t w, int32_t stride_d,
                               int32_t pad_d, int32_t *read_flags, int32_t n_frames, void *stream);
/* ... written as the reference's middle output instead: bev [frame][c * planes + d][h][w] (modules/voxelnet/Pipe.py:40-41, the
 * (C, D, H, W) result viewed as (C * D, H, W)); equal to mvx_bn_apply_tiles_frames followed by mvx_cl_to_bev_frames bit for bit,
 * without the channels-last tensor in between.  channels <= 64. */
int mvx_bn_apply_tiles_bev_frames(const float *y, const float *mean_inv, const float *c_bg, const int32_t *tile_flags, float *bev,
                                  int32_t planes, int32_t h, int32_t w, int32_t channels, int32_t n_frames, void *stream);
int mvx_conv3d_background_taps_frames(const float *w, const float *c_in, int32_t din, int32_t dout, int32_t cin, int32_t cout,
                                      int32_t stride_d, int32_t pad_d, float *bg, int32_t n_frames, void *stream);
int mvx_conv3d_background(const float *w, const float *c_in, int32_t din, int32_t dout, int32_t cin, int32_t cout,
                          int32_t stride_d, int32_t pad_d, float *bg_pre, void *stream);
int mvx_bn_background(const float *bg_pre, const float *bias, const float *mean_inv, int32_t planes, int32_t channels,
                      int32_t flags, float *y_bg, float *c_out, void *stream);
int mvx_conv3d_forward_bg(const float *in, const float *wpk, const float *bias, float *out, double *stats,
                          int32_t din, int32_t dout, int32_t h, int32_t w, int32_t cin, int32_t cout,
                          int32_t stride_d, int32_t pad_d, int32_t flags, const int32_t *in_halo_flags,
                          const uint8_t *out_mask, const float *bg_pre, int32_t border_active,
                          uint64_t *exec_stages, uint32_t *done_counter, double count, double eps, float *mean_inv,
                          uint32_t *work_counter, void *stream);
size_t mvx_conv3d_wgrad_bg_workspace_bytes(int32_t dout, int32_t h, int32_t w, int32_t cin, int32_t cout);
int mvx_conv3d_wgrad_bg(const float *in, const float *dz, float *dw, int32_t din, int32_t dout, int32_t h,
                        int32_t w, int32_t cin, int32_t cout, int32_t stride_d, int32_t pad_d, int32_t flags,
                        const int32_t *in_halo_flags, const float *c_in, const float *tap_sums, void *workspace,
                        size_t workspace_bytes, void *stream);
size_t mvx_plane_tap_sums_workspace_bytes(int32_t planes, int32_t channels);
int mvx_plane_tap_sums(const float *dz, int32_t planes, int32_t h, int32_t w, int32_t channels,
                       const int32_t *tile_flags, const float *inactive_sums, float *tap_sums, void *workspace,
                       size_t workspace_bytes, void *stream);
int mvx_tile_dilate_flags(const int32_t *in_tile_flags, const int32_t *self_tile_flags, int32_t din, int32_t dout,
                          int32_t h, int32_t w, int32_t stride_d, int32_t pad_d, int32_t *out_tile_flags, void *stream);
int mvx_conv3d_input_grad_sums(const float *w, const float *tap_sums, int32_t din, int32_t dout, int32_t cin,
                               int32_t cout, int32_t stride_d, int32_t pad_d, float *plane_grad_sums, void *stream);
int mvx_conv3d_dgrad_tiles(const float *dz, const float *wpk_dgrad, float *dx, int32_t din, int32_t dout, int32_t h,
                           int32_t w, int32_t cin, int32_t cout, int32_t stride_d, int32_t pad_d,
                           const int32_t *dx_tile_flags, uint64_t *exec_stages, uint32_t *work_counter, void *stream);
/* bf16x3 forms of the background-aware entry points (csrc/conv3d_split.hip; weights from mvx_conv3d_pack_weights_split) */
int mvx_conv3d_forward_bg_split(const float *in, const void *wsplit, const float *bias, float *out, double *stats,
                                int32_t din, int32_t dout, int32_t h, int32_t w, int32_t cin, int32_t cout,
                                int32_t stride_d, int32_t pad_d, int32_t flags, const int32_t *in_halo_flags,
                                const uint8_t *out_mask, const float *bg_pre, int32_t border_active, void *stream);
int mvx_conv3d_dgrad_tiles_split(const float *dz, const void *wsplit_dgrad, float *dx, int32_t din, int32_t dout,
                                 int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t stride_d, int32_t pad_d,
                                 int32_t flags, const int32_t *dx_tile_flags, void *stream);
size_t mvx_conv3d_wgrad_bg_split_workspace_bytes(int32_t dout, int32_t h, int32_t w, int32_t cin, int32_t cout);
int mvx_conv3d_wgrad_bg_split(const float *in, const float *dz, float *dw, int32_t din, int32_t dout, int32_t h,
                              int32_t w, int32_t cin, int32_t cout, int32_t stride_d, int32_t pad_d, int32_t flags,
                              const int32_t *in_halo_flags, const float *c_in, const float *tap_sums, void *workspace,
                              size_t workspace_bytes, void *stream);
/* ... and their frame-set forms (planes of n_frames frames stacked along depth, per-frame statistics [F][R][2][cout];
 * exec_stages (optional) u64 [1] += executed (depth tap, 32-channel chunk) stages, as in the f32 kernels) */
int mvx_conv3d_forward_bg_split_frames(const float *in, const void *wsplit, const float *bias, float *out, double *stats,
                                       int32_t din, int32_t dout, int32_t h, int32_t w, int32_t cin, int32_t cout,
                                       int32_t stride_d, int32_t pad_d, int32_t flags, const int32_t *in_halo_flags,
                                       const uint8_t *out_mask, const float *bg_pre, int32_t border_active,
                                       uint64_t *exec_stages, int32_t n_frames, void *stream);
int mvx_conv3d_dgrad_tiles_split_frames(const float *dz, const void *wsplit_dgrad, float *dx, int32_t din, int32_t dout,
                                        int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t stride_d, int32_t pad_d,
                                        int32_t flags, const int32_t *dx_tile_flags, uint64_t *exec_stages, int32_t n_frames,
                                        void *stream);
size_t mvx_conv3d_wgrad_bg_split_workspace_bytes_frames(int32_t dout, int32_t h, int32_t w, int32_t cin, int32_t cout,
                                                        int32_t n_frames);
int mvx_conv3d_wgrad_bg_split_frames(const float *in, const float *dz, float *dw, int32_t din, int32_t dout, int32_t h,
                                     int32_t w, int32_t cin, int32_t cout, int32_t stride_d, int32_t pad_d, int32_t flags,
                                     const int32_t *in_halo_flags, const float *c_in, const float *tap_sums,
                                     void *workspace, size_t workspace_bytes, int32_t n_frames, void *stream);
/* bf16x3 forms of the RPN's 2-D convolutions on frame sets (modules/voxelnet/Pipe.py:45-75; one plane per frame).  wsplit:
 * the 2-D kernel placed in the middle depth slice of a [cout][cin][3][3][3] tensor, packed by
 * mvx_conv3d_pack_weights_split; dw3 f32 [cout][cin][3][3][3] (overwritten): the 2-D gradient is its middle depth slice. */
int mvx_conv2d_forward_split_frames(const float *in, const void *wsplit, const float *bias, float *out, double *stats,
                                    int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t flags, int32_t n_frames,
                                    void *stream);
int mvx_conv2d_dgrad_split_frames(const float *dz, const void *wsplit_dgrad, float *dx, int32_t h, int32_t w, int32_t cin,
                                  int32_t cout, int32_t flags, int32_t n_frames, void *stream);
size_t mvx_conv2d_wgrad_split_workspace_bytes_frames(int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t n_frames);
int mvx_conv2d_wgrad_split_frames(const float *in, const float *dz, float *dw3, int32_t h, int32_t w, int32_t cin,
                                  int32_t cout, int32_t flags, void *workspace, size_t workspace_bytes, int32_t n_frames,
                                  void *stream);
size_t mvx_bn_relu_backward_tiles_workspace_bytes(int32_t planes, int32_t h, int32_t w, int32_t channels);
int mvx_bn_relu_backward_tiles(const float *dyhat, const float *y, const float *mean_inv, const float *c_bg,
                               const float *y_bg, const float *plane_grad_sums, const int32_t *tile_flags,
                               int32_t planes, int32_t h, int32_t w, int32_t channels, float *dz, float *dbias,
                               float *dz_inactive_sums, int32_t flags, void *workspace, size_t workspace_bytes,
                               void *stream);

/* ------------------------------------------------------------------------------------------
 * Row-wise fully connected layer on the matrix cores (fp32 MFMA).  Replaces the nn.Linear /
 * 1x1 nn.Conv2d GEMMs of FCN and CRB2d (modules/layers/Blocks.py:9,14,35,39) and their autograd.
 *   x f32 [rows][ldx] (k columns used), w f32 [n][ldw] (or [k][ldw] when w_transposed),
 *   y f32 [rows][ldy] (n columns written) = [ReLU](x w^T + bias)
 *   stats (optional, replicated f64 [R][2][n]): per-column (sum, sum of squares) of y weighted by row_w
 *   row_w (optional) f32 [rows]: multiplicity of each row in the reference's dense tensor
 *   splitk_workspace (optional, mvx_linear_splitk_workspace_bytes): lets a skinny product without
 *   epilogue (no bias/ReLU/stats, ldy == n, few output blocks, long k) be split along k into slabs
 *   that are summed in a fixed order -- fills the chip when rows x n alone cannot.
 * mvx_linear_wgrad: dw f32 [n][k] = dz^T x  (dz f32 [rows][lddz]).
 * mvx_linear_forward_bn (and the done_counter / count / eps / mean_inv arguments of mvx_conv3d_forward_bg): the
 *   workgroup that finishes last turns the statistics into mean_inv f32 [2][n] (= mvx_bn_finalize) inside the same
 *   launch; done_counter u32 [1] must be zero (cleared here unless MVX_FLAG_PREZEROED).
 */
size_t mvx_linear_splitk_workspace_bytes(int64_t rows, int32_t n);
int mvx_linear_forward_bn(const float *x, int32_t ldx, const float *w, int32_t ldw, int32_t w_transposed,
                          const float *bias, float *y, int32_t ldy, double *stats, const float *row_w, int64_t rows,
                          int32_t k, int32_t n, int32_t flags, uint32_t *done_counter, double count, double eps,
                          float *mean_inv, void *stream);
int mvx_linear_forward(const float *x, int32_t ldx, const float *w, int32_t ldw, int32_t w_transposed,
                       const float *bias, float *y, int32_t ldy, double *stats, const float *row_w,
                       int64_t rows, int32_t k, int32_t n, int32_t flags, void *splitk_workspace,
                       size_t splitk_workspace_bytes, void *stream);
size_t mvx_linear_wgrad_workspace_bytes(int64_t rows, int32_t k, int32_t n);

/* ------------------------------------------------------------------------------------------
 * Row GEMMs on PRE-CUT operands (csrc/rowgemm_pre.hip): the same nn.Linear / 1x1 Conv2d products
 * (modules/layers/Blocks.py:9,14,35,39; the fusion MLP of modules/imhead/Pipe.py:84-104) in the split arithmetics
 * MVX_FLAG_SPLIT3 (three bf16 pieces = the f32 operand exactly, six MFMAs per product) / MVX_FLAG_SPLIT_F16 (two fp16 pieces),
 * with every operand stored as PLANES of 16-bit pieces -- u16 [pieces][rows][k], plane stride rows * k -- by the kernel that
 * produced it, so that the GEMM moves its tiles global -> LDS by DMA and issues nothing but matrix instructions and LDS reads.
 *   mvx_split_planes_bytes     size of the planes of a [rows][k] f32 matrix
 *   mvx_split_rows             f32 [rows][ldx] -> planes (weights, once per optimizer step; any tensor whose producer does not
 *                              write planes itself); `scale` multiplies the values first (fp16 pieces only; a power of two)
 *   mvx_linear_forward_pre_frames  y f32 [rows][ldy] = [ReLU](out_scale * a b^T + bias) from a = planes of x [rows][k] and
 *                              b = planes of the weight [n][k] (input gradient: of the transposed weight), per-frame
 *                              BatchNorm sums / finalisation exactly as mvx_linear_forward_bn_frames; k % 32 == 0
 *   mvx_linear_wgrad_pre       dw f32 [n][k] (+= with MVX_FLAG_ACCUMULATE) = out_scale * dz^T x from the planes of dz [rows][n]
 *                              and x [rows][k]; n % 128 == 0, k % 128 == 0; workspace: mvx_linear_wgrad_pre_workspace_bytes
 */
size_t mvx_split_planes_bytes(int64_t rows, int32_t k, int32_t flags);
int mvx_split_rows(const float *x, int32_t ldx, int64_t rows, int32_t k, void *planes, int32_t flags, float scale, void *stream);
int mvx_linear_forward_pre_frames(const void *a_planes, const void *b_planes, const float *bias, float *y, int32_t ldy,
                                  double *stats, const float *row_w, int64_t rows, int32_t k, int32_t n, int32_t flags,
                                  float out_scale, uint32_t *done_counter, double eps, float *mean_inv,
                                  const mvx_frames_t *frames_host, int32_t row_kind, void *stream);
size_t mvx_linear_wgrad_pre_workspace_bytes(int64_t rows, int32_t k, int32_t n);
int mvx_linear_wgrad_pre(const void *x_planes, const void *dz_planes, float *dw, int64_t rows, int32_t k, int32_t n,
                         int32_t flags, float out_scale, void *workspace, size_t workspace_bytes, void *stream);
/* ... over the rows [row_lo, row_hi) of planes holding plane_rows rows each (workspace of the whole range suffices): the weight
 * gradient of a row range whose dz planes are already written (mvx_bn_relu_backward_planes_part_frames). */
int mvx_linear_wgrad_pre_rows(const void *x_planes, const void *dz_planes, float *dw, int64_t plane_rows, int64_t row_lo,
                              int64_t row_hi, int32_t k, int32_t n, int32_t flags, float out_scale, void *workspace,
                              size_t workspace_bytes, void *stream);
int mvx_linear_wgrad(const float *x, int32_t ldx, const float *dz, int32_t lddz, float *dw, int64_t rows,
                     int32_t k, int32_t n, int32_t flags, void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * VFE glue on [n_voxels][t][channels] rows.  Replaces modules/voxelnet/Pipe.py:14-18
 * (BN -> max over t -> repeat -> concat) and modules/voxelnet/VoxelNet.py:30 (max over t),
 * with their autograd.  The max covers all t rows, padded ones included (no mask).
 *   mvx_vfe_bn_max_concat        out [V][t][2C] = [BN(y), max_t BN(y)], argmax i32 [V][C]
 *   mvx_vfe_max_concat_backward  dyhat [V][t][C] from grad_out [V][t][2C]
 *   mvx_bn_segment_max           out [V][C] = max_t BN(y), argmax i32 [V][C]
 *   mvx_segment_max_backward     dyhat [V][t][C] from dfeat [V][C]
 */
int mvx_vfe_bn_max_concat(const float *y, const float *mean_inv, float *out, int32_t *argmax,
                          int32_t n_voxels, int32_t t, int32_t channels, const int32_t *voff,
                          const int32_t *vcnt, int32_t n_real, void *stream);
int mvx_vfe_max_concat_backward(const float *grad_out, const int32_t *argmax, float *dyhat,
                                int32_t n_voxels, int32_t t, int32_t channels, const int32_t *voff,
                                const int32_t *vcnt, int32_t n_real, void *stream);
int mvx_bn_segment_max(const float *y, const float *mean_inv, float *out, int32_t *argmax,
                       int32_t n_voxels, int32_t t, int32_t channels, const int32_t *voff,
                       const int32_t *vcnt, int32_t n_real, void *stream);
int mvx_segment_max_backward(const float *dfeat, const int32_t *argmax, float *dyhat, int32_t n_voxels,
                             int32_t t, int32_t channels, const int32_t *voff, const int32_t *vcnt,
                             int32_t n_real, void *stream);

/* Compact rows: inside MVXNet all padded rows of a voxel are identical, so the t rows
 * of voxel v are stored as its vcnt[v] real rows (matrix rows voff[v] ...) plus ONE padded row
 * (matrix row n_real + v) that stands for the t - vcnt[v] identical padded rows: weight
 * row_w = t - vcnt[v] in the BatchNorm sums, gradient = sum over the rows it stands for.  The four
 * entry points above take voff/vcnt/n_real (NULL/NULL/0 = the dense [V][t][C] layout).
 *   mvx_voxel_row_offsets          row_map (mvx_row_compact_map) -> voff, vcnt i32 [V], row_w f32 [n_real+V]
 *   mvx_vfe_compact_input          VFE-1 input rows [n_real+V][7+F]: real row j = [voxels[rows_sel[j]][0:7],
 *                                  imfeat[j]], padded rows = [0 x 7, imfeat[n_real]]  (MVXNet.py:26)
 *   mvx_vfe_compact_input_backward dimfeat [n_real+1][F] from grad_out [n_real+V][7+F]; scratch f64 [F]
 */
int mvx_voxel_row_offsets(const int32_t *row_map, int32_t n_voxels, int32_t t, int32_t n_real,
                          int32_t *voff, int32_t *vcnt, float *row_w, void *stream);
int mvx_vfe_compact_input(const float *voxels, int32_t vox_channels, const int32_t *rows_sel,
                          const float *imfeat, int32_t feat_channels, int32_t n_real, int32_t n_voxels,
                          float *out, void *stream);
int mvx_vfe_compact_input_backward(const float *grad_out, int32_t feat_channels, int32_t n_real,
                                   int32_t n_voxels, float *dimfeat, double *scratch, void *stream);

/* ------------------------------------------------------------------------------------------
 * Point <-> image fusion sampling.  Replaces featureMaping (modules/imhead/Pipe.py:23-82).
 *   voxels   f32 [rows][vox_channels]  dense (V*T) voxel rows: xyz in columns 0..2, projected
 *            (row, col) in the LAST two columns; rows with x == y == z == 0 are padding and get
 *            all their columns zeroed IN PLACE (Pipe.py:54-59)
 *   feats_host[l]  device pointer of FPN level l, channels-last f32 [h_l][w_l][channels];
 *            feat_hw_host = {h_0, w_0, h_1, w_1, ...} (both arrays live on the HOST)
 *   out      f32 [rows or n_real(+1)][n_levels*channels]: level l in columns [l*C, (l+1)*C)
 *   row_map  NULL: dense output, padded rows written as zeros (Pipe.py:80);
 *            else i32 [rows] from mvx_row_compact_map: only real rows are sampled, into their
 *            compact row; the caller keeps one shared zero row for all padded rows
 *   status   bit0 = a sample index left the (zero-padded) map: the reference's assert, Pipe.py:71
 * mvx_row_compact_map: row_map[r] = rank of real row r among real rows, -1 for padded rows;
 *            rows_sel (optional) i32 [rows]: inverse list; n_real i32 [1] on the device.  Padding rows
 *            (x == y == z == 0) get their channels 3.. zeroed IN PLACE here (Pipe.py:54-59), so that
 * mvx_feature_sample_rows, the compact sampler, only visits the n_real real rows (dense row rows_sel[j]
 *            -> out row j); same arithmetic and status as mvx_feature_sample.
 * mvx_expand_rows / _backward: compact [n][C] <-> dense [rows][C]; every padded row reads the
 *            shared row `pad_row`, whose gradient is the sum over the padded rows (C <= 256;
 *            scratch f64 [C]).
 */
size_t mvx_row_compact_workspace_bytes(int64_t rows);
int mvx_row_compact_map(float *voxels, int32_t vox_channels, int64_t rows, int32_t *row_map,
                        int32_t *rows_sel, int32_t *n_real, void *workspace, size_t workspace_bytes,
                        void *stream);
int mvx_feature_sample(float *voxels, int32_t vox_channels, int64_t rows, const int32_t *row_map,
                       const float *const *feats_host, const int32_t *feat_hw_host, int32_t n_levels,
                       int32_t channels, float imsize_h, float imsize_w, float eps, float *out,
                       int32_t *status, void *stream);
int mvx_feature_sample_rows(const float *voxels, int32_t vox_channels, const int32_t *rows_sel, int32_t n_real,
                            const float *const *feats_host, const int32_t *feat_hw_host, int32_t n_levels,
                            int32_t channels, float imsize_h, float imsize_w, float eps, float *out,
                            int32_t *status, void *stream);
int mvx_expand_rows(const float *compact, const int32_t *row_map, int32_t pad_row, float *out,
                    int64_t rows, int32_t channels, void *stream);
int mvx_expand_rows_backward(const float *grad_out, const int32_t *row_map, int32_t pad_row,
                             float *dcompact, double *scratch, int64_t rows, int32_t channels,
                             void *stream);

/* ------------------------------------------------------------------------------------------
 * Range crop, camera-frustum crop, lidar -> image projection.  Replaces
 * modules/data/Preprocessing.py:12-24 (crop / cropTensor), :26-55 (cropToSight) and
 * modules/utils/Calib.py:47-69 (lidar2Img).
 *
 * mvx_crop_points: order-preserving compaction of the points that pass the enabled filters.
 *   pcd f32 [F][cap_points][ncol], n_in i32 [F] (device, NULL = all cap_points live)
 *   range6_host  (lo xyz, hi xyz) f64 on the HOST or NULL: keep lo <= xyz < hi; bounds_f32 rounds
 *                the bounds to f32 first (cropTensor semantics)
 *   cam_from_velo_host = R0_rect @ Tr_velo_to_cam and p2_host = P2, row-major 4x4 f64 on the HOST,
 *                or NULL: keep cam.z > 0 and 0 <= (u, v) < (imsize_w, imsize_h) - 1e-3;
 *                math_f32 selects the torch-path (f32) arithmetic instead of the numpy-path (f64)
 *   out f32 [F][cap_points][ncol], n_out i32 [F]; src_index (optional) i32 [F][cap_points]
 * mvx_lidar2img: out[i][col_offset + 0..1] = (u, v), or (v, u) = (row, col) when swap_to_row_col
 *   (train.py:33); cam_z (optional) f32 [n] = depth in the camera frame.
 */
size_t mvx_crop_workspace_bytes(int32_t n_frames, int32_t cap_points);
int mvx_crop_points(const float *pcd, const int32_t *n_in, int32_t n_frames, int32_t cap_points,
                    int32_t ncol, const double *range6_host, int32_t bounds_f32,
                    const double *cam_from_velo_host, const double *p2_host, double imsize_w,
                    double imsize_h, int32_t math_f32, float *out, int32_t *n_out, int32_t *src_index,
                    void *workspace, size_t workspace_bytes, void *stream);
/* mvx_crop_project_points: the per-frame input preparation of the training loop in one compaction pass --
 *   crop + cropToSight with the numpy-path (f64) masks of cropdata.py:30-65, then for every kept point the f32
 *   projection of train.py:31-34 (lidar2Img on the torch path, swapped to (row, col)) appended as two columns:
 *   out f32 [F][cap_out][ncol + 2] = [x y z r ... row col], n_out i32 [F].  cam_from_velo / p2: the f64 matrices of the
 *   masks; proj_*: the matrices of the f32 projection (R0_rect @ Tr_velo_to_cam formed in f32, as torch does). */
size_t mvx_crop_project_workspace_bytes(int32_t n_frames, int32_t cap_points);
int mvx_crop_project_points(const float *pcd, const int32_t *n_in, int32_t n_frames, int32_t cap_points, int32_t ncol,
                            const double *range6_host, const double *cam_from_velo_host, const double *p2_host,
                            double imsize_w, double imsize_h, const double *proj_cam_from_velo_host,
                            const double *proj_p2_host, float *out, int32_t cap_out, int32_t *n_out, void *workspace,
                            size_t workspace_bytes, void *stream);
int mvx_lidar2img(const float *pcd, int32_t ncol, int64_t n_points, const double *cam_from_velo_host,
                  const double *p2_host, int32_t math_f32, float *out, int32_t ld_out, int32_t col_offset,
                  int32_t swap_to_row_col, float *cam_z, void *stream);

/* ------------------------------------------------------------------------------------------
 * Input-sparse first convolution as voxel GEMMs + index-grid gathers (exact up to fp32 summation
 * order).  Replaces VoxelNet.reindex + cml.conv1 (VoxelNet.py:16-22, Pipe.py:36) and their autograd
 * without ever building the 721 MB dense input grid:
 *   mvx_index_grid            grid i32 [d][h][w] = voxel id at occupied sites, -1 elsewhere, followed in the
 *                             same buffer by a coarse per-tile occupancy count (buffer size:
 *                             mvx_index_grid_bytes)
 *   forward:  P = X W_all^T   via mvx_linear_forward (X f32 [V][cin], W_all f32 [27*cout][cin],
 *                             row (kd*9+a*3+b)*cout + co = W[co][:][kd][a][b])
 *             mvx_sparse_conv_output:  out [dout][h][w][cout] = [ReLU](bias + sum of the P rows of the
 *                             <= 27 source voxels of each site), stats as in mvx_conv3d_forward
 *   backward: mvx_sparse_conv_gather_dz:  G f32 [V][27*cout], G[v][tap*cout + c] = dz at the output site
 *                             that read voxel v through tap (zero if none)
 *             dX = G W_all (mvx_linear_forward, w_transposed), dW_all = G^T X (mvx_linear_wgrad)
 */
size_t mvx_index_grid_bytes(int32_t d, int32_t h, int32_t w);
int mvx_index_grid(const int64_t *coords, int32_t n_voxels, int32_t d, int32_t h, int32_t w,
                   int32_t *grid, int32_t *status, void *stream);
int mvx_sparse_conv_output(const float *p, const int32_t *index_grid, const float *bias, float *out,
                           double *stats, int32_t din, int32_t dout, int32_t h, int32_t w, int32_t cout,
                           int32_t stride_d, int32_t pad_d, int32_t flags, void *stream);
int mvx_sparse_conv_gather_dz(const float *dz, const int64_t *coords, int32_t n_voxels, float *g_rows,
                              int32_t din, int32_t dout, int32_t h, int32_t w, int32_t cout,
                              int32_t stride_d, int32_t pad_d, void *stream);

/* ------------------------------------------------------------------------------------------
 * "bf16x3" variants of the dense convolution: every f32 operand is split into hi + lo bf16 and a
 * product is hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with f32 accumulation (per-product
 * relative error ~2e-5, i.e. fp32-grade for the 1e-4 feature bar; ~5x the rate of the exact-f32
 * MFMA).  Same arguments and layouts as the f32 entry points; weights are packed (and pre-split) by
 * mvx_conv3d_pack_weights_split into mvx_conv3d_packed_weight_bytes_split(cout, cin, flags) bytes.
 *
 * "bf16x6" (MVX_FLAG_SPLIT3 in `flags` of the pack AND of every call that uses the pack): THREE pieces per operand
 * (hi + mid + lo = the f32 value exactly) and the six products hh + hm + mh + hl + lh + mm; the dropped cross terms are
 * below 2^-25 of a product, i.e. under its f32 rounding: fp32-GRADE accuracy (tests hold it to the bounds of the
 * exact-f32 kernels) at 6/16 of the exact-f32 MFMA's matrix cycles.  Every `flags` below takes MVX_FLAG_SPLIT3.
 */
size_t mvx_conv3d_packed_weight_bytes_split(int32_t cout, int32_t cin, int32_t flags);
int mvx_conv3d_pack_weights_split(const float *w, void *wsplit, int32_t cout, int32_t cin, int32_t for_dgrad,
                                  int32_t flags, void *stream);
int mvx_conv3d_forward_split(const float *in, const void *wsplit, const float *bias, float *out, double *stats,
                             int32_t din, int32_t dout, int32_t h, int32_t w, int32_t cin, int32_t cout,
                             int32_t stride_d, int32_t pad_d, int32_t flags, void *stream);
int mvx_conv3d_dgrad_split(const float *dz, const void *wsplit_dgrad, float *dx, int32_t din, int32_t dout,
                           int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t stride_d, int32_t pad_d,
                           int32_t flags, void *stream);
int mvx_conv3d_wgrad_split(const float *in, const float *dz, float *dw, int32_t din, int32_t dout, int32_t h,
                           int32_t w, int32_t cin, int32_t cout, int32_t stride_d, int32_t pad_d, int32_t flags,
                           void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * Frame-set forms (see mvx_frames_t above): the same operations as the entry points of the same name without `_frames`,
 * for all frames of a step in ONE launch.  Conventions:
 *   - `frames_host` is a HOST pointer (copied by value into the launch); NULL = one frame;
 *   - `n_frames` forms: grids hold the frames stacked along depth -- [n_frames * planes][h][w][c]; din / dout / planes
 *     are PER FRAME; per-plane arrays (background constants, halo / tile flags, tap sums, plane gradient sums) are
 *     indexed by the global plane frame * planes + local plane;
 *   - per-frame results get a leading frame dimension: stats f64 [F][R][2][C], mean_inv f32 [F][2][C];
 *   - bias / weight gradients are summed over the frames (the reference accumulates them over the frames of a step).
 * Row forms take `row_kind` (MVX_ROWS_*).  BatchNorm populations per frame follow from the descriptor
 * (voxels of the frame x t), or, for MVX_ROWS_GRID, rows / n_frames.
 */
int mvx_linear_forward_bn_frames(const float *x, int32_t ldx, const float *w, int32_t ldw, int32_t w_transposed,
                                 const float *bias, float *y, int32_t ldy, double *stats, const float *row_w, int64_t rows,
                                 int32_t k, int32_t n, int32_t flags, uint32_t *done_counter, double eps, float *mean_inv,
                                 const mvx_frames_t *frames_host, int32_t row_kind, void *stream);
int mvx_bn_finalize_frames(const double *stats, double count, double eps, float *mean_inv, int32_t channels,
                           int32_t n_frames, void *stream);
int mvx_bn_apply_frames(const float *y, const float *mean_inv, float *out, int64_t rows, int32_t channels,
                        const mvx_frames_t *frames_host, int32_t row_kind, void *stream);
size_t mvx_bn_backward_scratch_bytes_frames(int32_t channels, int32_t n_frames);
int mvx_bn_relu_backward_frames(const float *dyhat, const float *y, const float *mean_inv, double count, float *dz,
                                float *dbias, double *scratch, const float *row_w, int64_t rows, int32_t channels,
                                int32_t flags, const mvx_frames_t *frames_host, int32_t row_kind,
                                float *dz_amax /* NULL or 1 float: max |dz| written (mvx_split_operand_amax) */, void *stream);
/* ... with dz written as three planes of bf16 pieces, u16 [3][rows][channels] (hi + mid + lo = dz exactly: the operand format
 * of mvx_linear_wgrad_pre), instead of f32 -- for a layer whose dz only feeds its own weight gradient (modules/imhead/Pipe.py:94:
 * the first fusion layer's input carries no gradient). */
int mvx_bn_relu_backward_planes_frames(const float *dyhat, const float *y, const float *mean_inv, double count, void *dz_planes,
                                       float *dbias, double *scratch, const float *row_w, int64_t rows, int32_t channels,
                                       int32_t flags, const mvx_frames_t *frames_host, int32_t row_kind, float *dz_amax,
                                       void *stream);
/* ... enqueued in nparts calls (part = 0 .. nparts - 1 in this order, same arguments and stream): part 0 runs the reduction pass
 * over all rows and the apply pass of the first row range, part p > 0 the apply pass of its range; part_rows[0..1] (host) receives
 * the rows [lo, hi) whose planes this call writes (the ranges tile [0, rows)).  The step's last weight gradient then runs range by
 * range (mvx_linear_wgrad_pre_rows, side stream) beside the apply pass of the next range instead of after the whole pass.  The bias
 * gradient is complete after the last part. */
int mvx_bn_relu_backward_planes_part_frames(const float *dyhat, const float *y, const float *mean_inv, double count,
                                            void *dz_planes, float *dbias, double *scratch, const float *row_w, int64_t rows,
                                            int32_t channels, int32_t flags, const mvx_frames_t *frames_host, int32_t row_kind,
                                            int32_t part, int32_t nparts, int64_t *part_rows, void *stream);
int mvx_vfe_bn_max_concat_frames(const float *y, const float *mean_inv, float *out, int32_t *argmax, int32_t n_voxels,
                                 int32_t t, int32_t channels, const int32_t *voff, const int32_t *vcnt, int32_t n_real,
                                 const mvx_frames_t *frames_host, void *stream);
int mvx_bn_segment_max_frames(const float *y, const float *mean_inv, float *out, int32_t *argmax, int32_t n_voxels,
                              int32_t t, int32_t channels, const int32_t *voff, const int32_t *vcnt, int32_t n_real,
                              const mvx_frames_t *frames_host, void *stream);
/* fusion_row_w (optional) f32 [n_real + n_frames]: row weights of the MVX_ROWS_FUSION layout (1 for real rows, the number
 * of padded rows of frame f for its shared padded row) */
int mvx_voxel_row_offsets_frames(const int32_t *row_map, int32_t n_voxels, int32_t t, int32_t n_real, int32_t *voff,
                                 int32_t *vcnt, float *row_w, float *fusion_row_w, const mvx_frames_t *frames_host,
                                 void *stream);
/* imfeat / dimfeat: MVX_ROWS_FUSION layout [n_real + n_frames][F]; scratch f64 [n_frames][F] */
int mvx_vfe_compact_input_frames(const float *voxels, int32_t vox_channels, const int32_t *rows_sel, const float *imfeat,
                                 int32_t feat_channels, int32_t n_real, int32_t n_voxels, float *out,
                                 const mvx_frames_t *frames_host, void *stream);
/* ... with a row pitch ld >= 7 + feat_channels (out [n_real + n_voxels][ld], the extra columns zero): a pitch that is a multiple of
 * 4 floats lets mvx_linear_forward* / mvx_linear_wgrad read the rows with 16-byte loads (k = ld, the weight padded with zero columns) */
int mvx_vfe_compact_input_pitch_frames(const float *voxels, int32_t vox_channels, const int32_t *rows_sel, const float *imfeat,
                                       int32_t feat_channels, int32_t n_real, int32_t n_voxels, float *out, int32_t ld,
                                       const mvx_frames_t *frames_host, void *stream);
int mvx_vfe_compact_input_backward_frames(const float *grad_out, int32_t feat_channels, int32_t n_real, int32_t n_voxels,
                                          float *dimfeat, double *scratch, const mvx_frames_t *frames_host, void *stream);
/* voxels of all frames back to back ([vox_off[F]][t][vox_channels]); real_off i32 [n_frames + 1] on the DEVICE receives the
 * real-row offsets of the frames (only vox_off and t of the descriptor are read) */
int mvx_row_compact_map_frames(float *voxels, int32_t vox_channels, int64_t rows, int32_t *row_map, int32_t *rows_sel,
                               int32_t *n_real, void *workspace, size_t workspace_bytes, const mvx_frames_t *frames_host,
                               int32_t *real_off, void *stream);
/* feats_host[f * n_levels + l] = level l of frame f (every frame has the same level shapes) */
int mvx_feature_sample_rows_frames(const float *voxels, int32_t vox_channels, const int32_t *rows_sel, int32_t n_real,
                                   const float *const *feats_host, const int32_t *feat_hw_host, int32_t n_levels,
                                   int32_t channels, float imsize_h, float imsize_w, float eps, float *out, int32_t *status,
                                   const mvx_frames_t *frames_host,
                                   float *out_amax /* NULL, or 1 float RAISED to max |out| (zero it first): mvx_split_operand_amax */, void *stream);
/* ... which also writes the rows as planes of bf16 pieces, u16 [3][plane_rows][n_levels * channels] (rows [0, n_real): hi + mid +
 * lo = the f32 value exactly; the caller clears the other rows): what mvx_split_rows would make of `out`, without reading it back.
 * out may be NULL: planes only (both readers of the rows -- mvx_linear_forward_pre_frames, mvx_linear_wgrad_pre -- take planes) */
int mvx_feature_sample_rows_planes_frames(const float *voxels, int32_t vox_channels, const int32_t *rows_sel, int32_t n_real,
                                          const float *const *feats_host, const int32_t *feat_hw_host, int32_t n_levels,
                                          int32_t channels, float imsize_h, float imsize_w, float eps, float *out, int32_t *status,
                                          const mvx_frames_t *frames_host, float *out_amax, void *planes, int64_t plane_rows,
                                          void *stream);
/* coords of all frames back to back; the frame of voxel v follows from vox_off (coords[:,0] is not read) */
size_t mvx_index_grid_bytes_frames(int32_t d, int32_t h, int32_t w, int32_t n_frames);
int mvx_index_grid_frames(const int64_t *coords, int32_t n_voxels, int32_t d, int32_t h, int32_t w, int32_t *grid,
                          int32_t *status, const mvx_frames_t *frames_host, void *stream);
int mvx_sparse_conv_output_frames(const float *p, const int32_t *index_grid, const float *bias, float *out, double *stats,
                                  int32_t din, int32_t dout, int32_t h, int32_t w, int32_t cout, int32_t stride_d,
                                  int32_t pad_d, int32_t flags, int32_t n_frames, void *stream);
/* ... told which 8 x 16 tiles of the OUTPUT hold a site with a voxel under its taps (the tile flags mvx_activity_dilate_frames forms
 * from the index grid, [n_frames * dout][tiles]): only those are built, the rest is ReLU(bias) (written, or implied under
 * MVX_FLAG_NO_BG_FILL); the BatchNorm sums count the rest in closed form as before. */
int mvx_sparse_conv_output_tiles_frames(const float *p, const int32_t *index_grid, const float *bias, float *out, double *stats,
                                        int32_t din, int32_t dout, int32_t h, int32_t w, int32_t cout, int32_t stride_d,
                                        int32_t pad_d, int32_t flags, int32_t n_frames, const int32_t *tile_flags, void *stream);
int mvx_sparse_conv_gather_dz_frames(const float *dz, const int64_t *coords, int32_t n_voxels, float *g_rows, int32_t din,
                                     int32_t dout, int32_t h, int32_t w, int32_t cout, int32_t stride_d, int32_t pad_d,
                                     const mvx_frames_t *frames_host, void *stream);
int mvx_activity_dilate_frames(const void *src, int32_t src_is_index, int32_t din, int32_t dout, int32_t h, int32_t w,
                               int32_t stride_d, int32_t pad_d, int32_t mark_border, uint8_t *dst_mask,
                               int32_t *dst_halo_flags, int32_t *dst_tile_flags, int32_t n_frames, void *stream);
int mvx_tile_dilate_flags_frames(const int32_t *in_tile_flags, const int32_t *self_tile_flags, int32_t din, int32_t dout,
                                 int32_t h, int32_t w, int32_t stride_d, int32_t pad_d, int32_t *out_tile_flags,
                                 int32_t n_frames, void *stream);
int mvx_conv3d_background_frames(const float *w, const float *c_in, int32_t din, int32_t dout, int32_t cin, int32_t cout,
                                 int32_t stride_d, int32_t pad_d, float *bg_pre, int32_t n_frames, void *stream);
int mvx_bn_background_frames(const float *bg_pre, const float *bias, const float *mean_inv, int32_t planes, int32_t channels,
                             int32_t flags, float *y_bg, float *c_out, int32_t n_frames, void *stream);
int mvx_conv3d_forward_bg_frames(const float *in, const float *wpk, const float *bias, float *out, double *stats,
                                 int32_t din, int32_t dout, int32_t h, int32_t w, int32_t cin, int32_t cout,
                                 int32_t stride_d, int32_t pad_d, int32_t flags, const int32_t *in_halo_flags,
                                 const uint8_t *out_mask, const float *bg_pre, int32_t border_active, uint64_t *exec_stages,
                                 uint32_t *done_counter, double count, double eps, float *mean_inv, uint32_t *work_counter,
                                 int32_t n_frames, void *stream);
int mvx_conv3d_dgrad_tiles_frames(const float *dz, const float *wpk_dgrad, float *dx, int32_t din, int32_t dout, int32_t h,
                                  int32_t w, int32_t cin, int32_t cout, int32_t stride_d, int32_t pad_d,
                                  const int32_t *dx_tile_flags, uint64_t *exec_stages, uint32_t *work_counter,
                                  int32_t n_frames, void *stream);
int mvx_conv3d_input_grad_sums_frames(const float *w, const float *tap_sums, int32_t din, int32_t dout, int32_t cin,
                                      int32_t cout, int32_t stride_d, int32_t pad_d, float *plane_grad_sums,
                                      int32_t n_frames, void *stream);
size_t mvx_conv3d_wgrad_bg_workspace_bytes_frames(int32_t dout, int32_t h, int32_t w, int32_t cin, int32_t cout,
                                                  int32_t n_frames);
int mvx_conv3d_wgrad_bg_frames(const float *in, const float *dz, float *dw, int32_t din, int32_t dout, int32_t h, int32_t w,
                               int32_t cin, int32_t cout, int32_t stride_d, int32_t pad_d, int32_t flags,
                               const int32_t *in_halo_flags, const float *c_in, const float *tap_sums, void *workspace,
                               size_t workspace_bytes, int32_t n_frames, void *stream);
size_t mvx_bn_relu_backward_tiles_workspace_bytes_frames(int32_t planes, int32_t h, int32_t w, int32_t channels,
                                                         int32_t n_frames);
int mvx_bn_relu_backward_tiles_frames(const float *dyhat, const float *y, const float *mean_inv, const float *c_bg,
                                      const float *y_bg, const float *plane_grad_sums, const int32_t *tile_flags,
                                      int32_t planes, int32_t h, int32_t w, int32_t channels, float *dz, float *dbias,
                                      float *dz_inactive_sums, float *dz_amax /* NULL or 1 float: max |dz| written */,
                                      int32_t flags, void *workspace, size_t workspace_bytes, int32_t n_frames, void *stream);
/* cl f32 [n_frames * d][h][w][c]  <->  bev f32 [n_frames][c * d][h][w] */
int mvx_cl_to_bev_frames(const float *cl, float *bev, int32_t d, int32_t h, int32_t w, int32_t channels, int32_t reverse,
                         int32_t n_frames, void *stream);
/* one (C*D, H, W) map -> `copies` channels-last frames [copies][d][h][w][channels], one pass: a gradient of the middle output that
 * is shared by all frames of a step, in the layout and multiplicity the frame-set backward reads */
int mvx_bev_to_cl_broadcast(const float *bev, float *cl, int32_t d, int32_t h, int32_t w, int32_t channels, int32_t copies,
                            void *stream);

/* ------------------------------------------------------------------------------------------
 * Region proposal network on frame sets.  Replaces what RPN.forward and its autograd obtain from
 * ATen / MIOpen (modules/voxelnet/Pipe.py:45-75; CRB2d / DeCRB2d of modules/layers/Blocks.py:31-51): maps are
 * channels-last [n_frames][h][w][c] f32, BatchNorm statistics per frame.
 *
 *   mvx_conv2d_forward_frames  out = [ReLU](conv3x3(in) + bias), stride 1, padding 1; stats / in-kernel finalisation as
 *                              mvx_conv3d_forward_bg_frames (count = h * w per frame).  wpk from mvx_conv3d_pack_weights with
 *                              bit 1 of for_dgrad set (2-D source kernel).  MVX_FLAG_TAPS2: only taps {0,1}^2 carry weight:
 *                              a stride-2 3x3 convolution (padding 1) of an image X equals this 2x2-window convolution of
 *                              space_to_depth(X) with the weight rearranged as W2[co][(pr,pc,ci)][ta][tb] = W[co][ci][a][b],
 *                              a -> (ta, pr): 0 -> (0,1), 1 -> (1,0), 2 -> (1,1), likewise b -> (tb, pc).  Only 9 of the 16
 *                              (window tap, parity) blocks of W2 hold a kernel tap; the other 7 are zero BY THIS DEFINITION
 *                              and the three kernels do not execute them (forward / dgrad / wgrad run the true 9-tap FLOPs,
 *                              the weight gradient of those blocks is returned as zero).
 *   mvx_conv2d_dgrad_frames    dx from dz (flipped window {1,2}^2 under MVX_FLAG_TAPS2)
 *   mvx_conv2d_wgrad_frames    dw f32 [cout][cin][3][3] summed over all frames; cin % 64 == 0, cout % 64 == 0
 *   mvx_space_to_depth_frames  in [F*planes][h][w][c] -> out [F][h/2][w/2][4][planes][c], channel block p = 2*(y&1) + (x&1)
 *                              (reverse != 0: the inverse, `in` is the space-to-depth tensor and `out` the full one)
 *   mvx_d2s_bn_apply_frames    t [F][h][w][s*s][c] (row-GEMM output of a ConvTranspose2d with kernel = stride = s, column
 *                              order (i, j, co)) -> out[f][y*s+i][x*s+j][col_offset + co] = (t - mean_f) * inv_f with rows of
 *                              ld_out floats (mean_inv NULL: copy); reverse != 0: t is WRITTEN from the out slice
 *   mvx_bn_apply_strided_frames  y [rows][c] dense -> out rows of ld_out floats at col_offset (frames = equal row shares);
 *                              reverse != 0: y is WRITTEN from the out slice
 *   mvx_row_stats_frames       stats f64 [F][R][2][c] of y [F][rows / F][c]
 */
int mvx_conv2d_forward_frames(const float *in, const float *wpk, const float *bias, float *out, double *stats, int32_t h,
                              int32_t w, int32_t cin, int32_t cout, int32_t flags, uint32_t *done_counter, double eps,
                              float *mean_inv, uint32_t *work_counter, int32_t n_frames, void *stream);
int mvx_conv2d_dgrad_frames(const float *dz, const float *wpk_dgrad, float *dx, int32_t h, int32_t w, int32_t cin,
                            int32_t cout, int32_t flags, uint32_t *work_counter, int32_t n_frames, void *stream);
size_t mvx_conv2d_wgrad_workspace_bytes_frames(int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t n_frames);
int mvx_conv2d_wgrad_frames(const float *in, const float *dz, float *dw, int32_t h, int32_t w, int32_t cin, int32_t cout,
                            int32_t flags, void *workspace, size_t workspace_bytes, int32_t n_frames, void *stream);
int mvx_space_to_depth_frames(const float *in, float *out, int32_t n_frames, int32_t planes, int32_t h, int32_t w,
                              int32_t channels, int32_t reverse, void *stream);
int mvx_d2s_bn_apply_frames(float *t, const float *mean_inv, float *out, int32_t n_frames, int32_t h, int32_t w, int32_t s,
                            int32_t channels, int32_t ld_out, int32_t col_offset, int32_t reverse, void *stream);
int mvx_bn_apply_strided_frames(float *y, const float *mean_inv, float *out, int64_t rows, int32_t channels, int32_t ld_out,
                                int32_t col_offset, int32_t n_frames, int32_t reverse, void *stream);
int mvx_row_stats_frames(const float *y, double *stats, int64_t rows, int32_t channels, int32_t n_frames, void *stream);

/* ------------------------------------------------------------------------------------------
 * Target assignment and loss.
 *
 * mvx_bbox_pairwise: cpp.bboxOverlap / cpp.bboxIntersection (cpp/voxelutil.cpp:96-136; callers
 *   modules/augment/Augment.py:54).  boxes f32 [n][4][2] BEV corner points -> out f32 [n1][n2]: IoU (want_iou != 0)
 *   or intersection area, with the reference's origin-fan clipping arithmetic in f32.  The reference fills r2[j]
 *   (box index) instead of r2[k] (corner index) at :107-109 -- out of bounds beyond 5 boxes; computed here is what its
 *   callers expect, box i against box j.
 *
 * mvx_classify_anchors: cpp._classifyAnchors (cpp/voxelutil.cpp:138-316; modules/Calc.py:88-96; train.py:46).
 *   gts f32 [n_gt][4][2], anchors f32 [l][w][anchors_per_loc][4][2] (BEV corners), nls / nws i64 [n_gt] = centre cell of
 *   every ground truth (Calc.py:91-94).  Output, in the reference's visiting order (ground truth, orientation, rows
 *   up then down, columns right then left):
 *     pos_idx i64 [3][cap] (x, y, z of the positives: IoU >= pos_thr), gi i64 [cap] their ground-truth ids,
 *     neg_idx i64 [3][cap] (the NOT-negative anchors: IoU >= neg_thr, positives included), counts i32 [2] on the device.
 *   window_radius R (1..55): the walk is replayed inside a (2R+1)^2 window of cells around the centre; R must cover
 *   every cell with IoU >= 0.1 (half the sum of the two box diagonals / cell size).  status i32 [1] (OR-ed): 1 = the walk
 *   reached the window edge (raise R), 2 = cap exceeded, 4 = a centre cell outside the grid (ground truth skipped;
 *   the reference indexes out of bounds there).
 *   One documented difference: when the reference's crossing test fails on a near-degenerate edge (|s2 - s1| <= 1e-6,
 *   :44) its static scratch keeps a point from an EARLIER call; this library takes the edge's start point instead.
 * mvx_classify_anchors_frames: the same for the frames of a step in ONE walk launch (train.py:46 once per frame): gts / nls /
 *   nws hold the ground truths of all frames back to back, gt_off_host i32 [n_frames + 1] (HOST memory, read during the
 *   call) their offsets; frame f's lists go to pos_idx[f][3][cap], neg_idx[f][3][cap], gi[f][cap] (ground-truth ids local
 *   to the frame) and counts[f][2]; status is OR-ed over the frames.  workspace: the _workspace_bytes of the TOTAL count.
 *
 * mvx_voxel_loss: VoxelLoss forward + backward (modules/voxelnet/Loss.py:15-45; train.py:140,161).
 *   score f32 (l, w, anchors_per_loc) and reg f32 (l, w, 7*anchors_per_loc) through explicit element strides (the RPN's
 *   (1,2,L,W) / (1,14,L,W) maps are read in place); pos_idx / neg_idx / gi as above with leading dimensions pos_ld /
 *   neg_ld; gts f32 [n][gt_ld >= 7] xyzlwhr, anchors f32 [l][w][anchors_per_loc][7].
 *   losses f32 [2] = (clsLoss, regLoss); dscore / dreg (optional) = d(clsLoss)/dscore and d(regLoss)/dreg written through
 *   their own strides -- dreg must be ZERO on entry (only positive anchors receive a gradient; repeated anchors add up).
 *   n_pos = 0: regLoss = 0 (the reference returns None); n_pos = n_neg = 0 is the `pi is None` branch (Loss.py:17-19).
 *   counts_dev (optional) i32 [2]: read n_pos / n_neg from the device instead (forward only: dscore must be NULL).
 *   scratch f64 [4].
 */
int mvx_bbox_pairwise(const float *boxes1, int32_t n1, const float *boxes2, int32_t n2, int32_t want_iou, float *out,
                      void *stream);
size_t mvx_classify_anchors_workspace_bytes(int32_t n_gt, int32_t anchors_per_loc, int32_t window_radius);
int mvx_classify_anchors(const float *gts, int32_t n_gt, const float *anchors, int32_t l, int32_t w,
                         int32_t anchors_per_loc, const int64_t *nls, const int64_t *nws, float neg_thr, float pos_thr,
                         int32_t window_radius, int64_t *pos_idx, int64_t *neg_idx, int64_t *gi, int64_t cap,
                         int32_t *counts, int32_t *status, void *workspace, size_t workspace_bytes, void *stream);
int mvx_classify_anchors_frames(const float *gts, const int32_t *gt_off_host, int32_t n_frames, const float *anchors, int32_t l,
                                int32_t w, int32_t anchors_per_loc, const int64_t *nls, const int64_t *nws, float neg_thr,
                                float pos_thr, int32_t window_radius, int64_t *pos_idx, int64_t *neg_idx, int64_t *gi,
                                int64_t cap, int32_t *counts, int32_t *status, void *workspace, size_t workspace_bytes,
                                void *stream);
int mvx_voxel_loss(const float *score, int64_t score_sl, int64_t score_sw, int64_t score_sa, const float *reg,
                   int64_t reg_sl, int64_t reg_sw, int64_t reg_sc, const int64_t *pos_idx, int64_t pos_ld,
                   const int64_t *neg_idx, int64_t neg_ld, const int64_t *gi, const int32_t *counts_dev, int32_t n_pos,
                   int32_t n_neg, const float *gts, int32_t gt_ld, const float *anchors, int32_t l, int32_t w,
                   int32_t anchors_per_loc, float a, float b, float eps, float *dscore, int64_t dscore_sl,
                   int64_t dscore_sw, int64_t dscore_sa, float *dreg, int64_t dreg_sl, int64_t dreg_sw, int64_t dreg_sc,
                   float *losses, double *scratch, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MVX_HIP_H */
