// Activity maps and background constants of the CML convolution stack.
//
// The grid VoxelNet.reindex fills (modules/voxelnet/VoxelNet.py:16-22) is zero outside the V voxel
// sites, and every block of CML (modules/voxelnet/Pipe.py:31-43) is Conv3d -> ReLU -> BatchNorm
// without affine (modules/layers/Blocks.py:20-29).  A site whose whole receptive field holds no voxel
// therefore carries, after each layer, ONE value per channel and depth plane -- ReLU(bias) after the
// first convolution, ReLU(bias + sum_taps W * c_prev) after the next ones -- the same at every such
// "background" site, except where the 3x3 window leaves the image (zero padding is not the background).
// These kernels find the background sites (exact dilation of the voxel occupancy, layer by layer) and
// evaluate the constants, so that the convolution kernels can
//   - forward: fill background tiles with the constant instead of convolving them,
//   - wgrad  : sum (x - c) (x) dz over the tiles that hold a non-background site only and add the
//              remaining c (x) sum(dz) term in closed form,
// both exact rewrites of the dense arithmetic (differences at fp32 rounding level).
#include "common.h"

namespace {

constexpr int ATH = 8, ATW = 16;        // tile of the convolution kernels (conv3d.hip TH x TW)

// dst[d][y][x] = 1 iff any source site in the 3x3x3 receptive field is active, or (mark_border and the
// in-plane window leaves the image).  Source: int32 index grid (voxel id, -1 = empty) or uint8 mask.
__global__ void activity_sites(const void *__restrict__ src, int src_is_index, int Din, int Dout, int H, int W, int sd,
                               int pd, int mark_border, unsigned char *__restrict__ dst, int n_frames) {
    const size_t n = (size_t)n_frames * Dout * H * W;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(e % W), y = (int)((e / W) % H), d = (int)(e / ((size_t)W * H));
        int on = mark_border && (y == 0 || y == H - 1 || x == 0 || x == W - 1);
        for (int kd = 0; kd < 3 && !on; ++kd) {
            const int ds = mvx_src_plane(d, Din, Dout, sd, pd, kd);
            if (ds < 0) continue;
            for (int a = -1; a <= 1 && !on; ++a) {
                const int yy = y + a;
                if (yy < 0 || yy >= H) continue;
                for (int b = -1; b <= 1; ++b) {
                    const int xx = x + b;
                    if (xx < 0 || xx >= W) continue;
                    const size_t s = ((size_t)ds * H + yy) * W + xx;
                    on |= src_is_index ? (((const int *)src)[s] >= 0) : (((const unsigned char *)src)[s] != 0);
                }
            }
        }
        dst[e] = (unsigned char)on;
    }
}


// The same for four consecutive sites of a row per thread (W a multiple of 4, 16-byte / 4-byte aligned rows): each of the (up to)
// nine source rows is read as ONE 16-byte (index grid) or 4-byte (mask) vector plus its two neighbours -- 27 loads per four
// sites instead of 27 per site -- and the four results leave as one 32-bit store.
__global__ __launch_bounds__(256) void activity_sites4(const void *__restrict__ src, int src_is_index, int Din, int Dout, int H,
                                                       int W, int sd, int pd, int mark_border, unsigned char *__restrict__ dst,
                                                       int n_frames) {
    const int W4 = W >> 2;
    const size_t n = (size_t)n_frames * Dout * H * W4;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(e % W4) * 4, y = (int)((e / W4) % H), d = (int)(e / ((size_t)W4 * H));
        unsigned on = 0u;                                   // bit k: site x + k
        if (mark_border) {
            if (y == 0 || y == H - 1) on = 15u;
            if (x == 0) on |= 1u;
            if (x + 4 == W) on |= 8u;
        }
        for (int kd = 0; kd < 3 && on != 15u; ++kd) {
            const int ds = mvx_src_plane(d, Din, Dout, sd, pd, kd);
            if (ds < 0) continue;
            for (int a = -1; a <= 1 && on != 15u; ++a) {
                const int yy = y + a;
                if (yy < 0 || yy >= H) continue;
                const size_t s = ((size_t)ds * H + yy) * W + x;
                unsigned m;                                 // bit 0: x - 1, bits 1..4: x .. x + 3, bit 5: x + 4
                if (src_is_index) {
                    const int *row = (const int *)src + s;
                    const int4 v = *(const int4 *)row;
                    m = (v.x >= 0 ? 2u : 0u) | (v.y >= 0 ? 4u : 0u) | (v.z >= 0 ? 8u : 0u) | (v.w >= 0 ? 16u : 0u);
                    if (x > 0 && row[-1] >= 0) m |= 1u;
                    if (x + 4 < W && row[4] >= 0) m |= 32u;
                } else {
                    const unsigned char *row = (const unsigned char *)src + s;
                    const unsigned v = *(const unsigned *)row;
                    m = ((v & 0xffu) ? 2u : 0u) | ((v & 0xff00u) ? 4u : 0u) | ((v & 0xff0000u) ? 8u : 0u) | ((v & 0xff000000u) ? 16u : 0u);
                    if (x > 0 && row[-1]) m |= 1u;
                    if (x + 4 < W && row[4]) m |= 32u;
                }
                on |= (m | (m >> 1) | (m >> 2)) & 15u;      // site k sees bits k, k + 1, k + 2 of m
            }
        }
        *(unsigned *)(dst + e * 4) = ((on & 1u) ? 1u : 0u) | ((on & 2u) ? 0x100u : 0u) | ((on & 4u) ? 0x10000u : 0u) | ((on & 8u) ? 0x1000000u : 0u);
    }
}

// flags[d][tile] = 1 iff the (TH+2) x (TW+2) halo of the tile holds an active site of plane d
// tile_flags[d][tile] = 1 iff the tile itself (no halo) holds one
__global__ __launch_bounds__(256) void activity_halo_flags(const unsigned char *__restrict__ mask, int D, int H, int W,
                                                           int *__restrict__ flags, int *__restrict__ tile_flags) {
    const int tiles_x = (W + ATW - 1) / ATW;
    const int tx0 = (blockIdx.x % tiles_x) * ATW - 1, ty0 = (blockIdx.x / tiles_x) * ATH - 1;
    const int d = blockIdx.y;
    int on = 0, inner = 0;
    if (threadIdx.x < (ATH + 2) * (ATW + 2)) {
        const int hy = threadIdx.x / (ATW + 2), hx = threadIdx.x % (ATW + 2);
        const int y = ty0 + hy, x = tx0 + hx;
        if (y >= 0 && y < H && x >= 0 && x < W) on = mask[((size_t)d * H + y) * W + x];
        inner = on && hy >= 1 && hy <= ATH && hx >= 1 && hx <= ATW;
    }
    on = __syncthreads_or(on);
    inner = __syncthreads_or(inner);
    if (threadIdx.x == 0) {
        if (flags) flags[(size_t)d * gridDim.x + blockIdx.x] = on ? 1 : 0;
        if (tile_flags) tile_flags[(size_t)d * gridDim.x + blockIdx.x] = inner ? 1 : 0;
    }
}

// out[d][t] = self[d][t] | any 3x3 tile neighbour flagged in a plane of `in` that is tied to plane d by a depth tap:
// the tiles of this layer's OUTPUT GRADIENT that the consumers of its restricted backward read (the halo of a
// flagged input tile reaches into all 8 neighbours).  in: flags of the conv INPUT [Din], self/out: [Dout].
__global__ void tile_dilate_flags(const int *__restrict__ in, const int *__restrict__ self, int Din, int Dout, int tiles_y,
                                  int tiles_x, int sd, int pd, int *__restrict__ out, int n_frames) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int nt = tiles_y * tiles_x;
    if (e >= n_frames * Dout * nt) return;
    const int d = e / nt, t = e - d * nt, ty = t / tiles_x, tx = t - ty * tiles_x;
    int on = self ? self[e] : 0;
    for (int kd = 0; kd < 3 && !on; ++kd) {
        const int ds = mvx_src_plane(d, Din, Dout, sd, pd, kd);
        if (ds < 0) continue;
        for (int a = -1; a <= 1 && !on; ++a)
            for (int b = -1; b <= 1; ++b) {
                const int yy = ty + a, xx = tx + b;
                if (yy >= 0 && yy < tiles_y && xx >= 0 && xx < tiles_x) on |= in[(size_t)ds * nt + yy * tiles_x + xx];
            }
    }
    out[e] = on ? 1 : 0;
}

// bg_pre[d][n] = sum over the valid depth taps of plane d and all in-plane taps / channels of
//                W[n][c][kd][a][b] * c_in[src(d,kd)][c]            (f64 accumulation, rounded once)
// bg_tap (optional) [planes][3][Cout]: the same sum per depth tap (0 for a tap without a source plane) -- what a depth tap
// contributes to an interior output site when every source site of its window is background.
// bg_cls (optional) [planes][9][Cout]: the sum for an output site on the image border, class = 3 * ry + rx with
// ry = 0 / 1 / 2 for the first / an inner / the last image row (rx likewise for columns): in-plane taps that fall outside the
// image are dropped (zero padding), all source sites inside hold the background.
__global__ __launch_bounds__(64) void conv_background(const float *__restrict__ w, const float *__restrict__ c_in, int Din,
                                                      int Dout, int Cin, int Cout, int sd, int pd,
                                                      float *__restrict__ bg_pre, float *__restrict__ bg_tap,
                                                      float *__restrict__ bg_cls) {
    const int n = blockIdx.x, d = blockIdx.y;        // one wave per (output channel, global plane); lanes over (c, tap)
    double s = 0.0;
    double cls[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) cls[q] = 0.0;
    for (int kd = 0; kd < 3; ++kd) {
        const int ds = mvx_src_plane(d, Din, Dout, sd, pd, kd);
        double sk = 0.0;
        if (ds >= 0)
            for (int e = threadIdx.x; e < Cin * 9; e += 64) {
                const int c = e / 9, k = e - c * 9, a = k / 3, b = k - a * 3;
                const double t = (double)w[(((size_t)n * Cin + c) * 3 + kd) * 9 + k] * (double)c_in[(size_t)ds * Cin + c];
                sk += t;
                if (bg_cls) {
#pragma unroll
                    for (int q = 0; q < 9; ++q) {
                        const int ry = q / 3, rx = q - ry * 3;
                        // tap row a reads source row gy + a - 1: outside for the first image row when a == 0, for the last when a == 2
                        const bool inside = !(ry == 0 && a == 0) && !(ry == 2 && a == 2) && !(rx == 0 && b == 0) && !(rx == 2 && b == 2);
                        if (inside) cls[q] += t;
                    }
                }
            }
        s += sk;
        if (bg_tap) {
            sk = wave_sum_f64(sk);
            if (threadIdx.x == 0) bg_tap[((size_t)d * 3 + kd) * Cout + n] = (float)sk;
        }
    }
    s = wave_sum_f64(s);
    if (threadIdx.x == 0) bg_pre[(size_t)d * Cout + n] = (float)s;
    if (bg_cls) {
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const double v = wave_sum_f64(cls[q]);
            if (threadIdx.x == 0) bg_cls[((size_t)d * 9 + q) * Cout + n] = (float)v;
        }
    }
}

// y_bg = [ReLU](bg_pre + bias) and c_out = (y_bg - mean) * inv, with exactly the fp32 operations of the
// convolution epilogue and of bn_apply, so that c_out equals the normalised tensor at background sites bit for bit.
__global__ void bn_background(const float *__restrict__ bg_pre, const float *__restrict__ bias, const float *__restrict__ mi,
                              int D, int C, int relu, float *__restrict__ y_bg, float *__restrict__ c_out, int n_frames) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_frames * D * C) return;
    const int c = e % C;
    mi += (size_t)(e / (D * C)) * 2 * C;                  // the plane's frame
    float v = (bg_pre ? bg_pre[e] : 0.f) + (bias ? bias[c] : 0.f);
    if (relu) v = fmaxf(v, 0.f);
    if (y_bg) y_bg[e] = v;
    c_out[e] = (v - mi[c]) * mi[C + c];
}

// ---- BatchNorm apply with the background: out = (y - mean) * inv on the tiles that hold a non-background site, the
// plane's normalised background constant c_bg[plane][c] everywhere else WITHOUT reading y there (bit-identical to bn_apply:
// c_bg is formed with the same fp32 operations, see bn_background).  One workgroup per (tile, plane).
__global__ __launch_bounds__(256) void bn_apply_tiles(const float *__restrict__ y, const float *__restrict__ mi,
                                                      const float *__restrict__ c_bg, const int *__restrict__ tile_flags,
                                                      float *__restrict__ out, int D, int H, int W, int C, int ntiles,
                                                      const int *__restrict__ read_flags) {
    const int tiles_x = (W + ATW - 1) / ATW;
    const int t = blockIdx.x, d = blockIdx.y, frame = d / D;
    // read_flags (may be NULL): the tiles some consumer reads (tile_read_flags below); a background tile outside that set is not
    // written -- its constant would never be looked at
    if (read_flags && !read_flags[(size_t)d * ntiles + t] && !tile_flags[(size_t)d * ntiles + t]) return;
    const int c4n = C >> 2, ct = threadIdx.x % c4n, st = threadIdx.x / c4n, spb = 256 / c4n;
    const int ty0 = (t / tiles_x) * ATH, tx0 = (t % tiles_x) * ATW;
    const bool on = tile_flags[(size_t)d * ntiles + t] != 0;          // block-uniform
    const float4 cb = *(const float4 *)(c_bg + (size_t)d * C + ct * 4);
    const float *fmi = mi + (size_t)frame * 2 * C;
    const float4 m = *(const float4 *)(fmi + ct * 4), iv = *(const float4 *)(fmi + C + ct * 4);
    for (int sidx = st; sidx < ATH * ATW; sidx += spb) {
        const int gy = ty0 + sidx / ATW, gx = tx0 + sidx % ATW;
        if (gy >= H || gx >= W) continue;
        const size_t off = (((size_t)d * H + gy) * W + gx) * C + ct * 4;
        float4 o = cb;
        if (on) {
            const float4 v = *(const float4 *)(y + off);
            o = make_float4((v.x - m.x) * iv.x, (v.y - m.y) * iv.y, (v.z - m.z) * iv.z, (v.w - m.w) * iv.w);
        }
        *(float4 *)(out + off) = o;
    }
}

// ---- which tiles of a layer's INPUT does the background-aware gather of that layer read?  The gather computes output tile T of
// output plane d when T's unit lies on the image border or one of the unit's two tiles has a flagged source halo in some valid depth
// tap (conv3d_gather_splitT / conv3d_gather_pf: `active`; the 16 x 16-site units of the split kernels pair the tiles (tx, 2 k) and
// (tx, 2 k + 1)), and it then reads the 3 x 3 tile neighbourhood of T in every valid source plane.  The weight gradient's step list
// (flagged source halo per depth tap) is a subset of that.  read[p][t] = some computed (d, T) with p a source plane of d and
// t in T's neighbourhood: a SUPERSET of what is read (idle border tiles and skipped depth taps read less).  One thread per (p, t).
__global__ void tile_read_flags(const int *__restrict__ hflag, int Din, int Dout, int ty_n, int tx_n, int sd, int pd,
                                int *__restrict__ read, int n_frames) {
    const int ntiles = ty_n * tx_n;
    const long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (e >= (long long)n_frames * Din * ntiles) return;
    const int p = (int)(e / ntiles), t = (int)(e - (long long)p * ntiles);
    const int ty = t / tx_n, tx = t - ty * tx_n;
    int r = 0;
    for (int kd = 0; kd < 3 && !r; ++kd) {
        const int d = mvx_dst_plane(p, Din, Dout, sd, pd, kd);
        if (d < 0) continue;
        for (int y = max(ty - 1, 0); y <= min(ty + 1, ty_n - 1) && !r; ++y)
            for (int x = max(tx - 1, 0); x <= min(tx + 1, tx_n - 1) && !r; ++x) {
                const int y0 = y & ~1, y1 = min(y0 + 1, ty_n - 1);                    // the unit's two tiles
                const bool border = x == 0 || x == tx_n - 1 || y0 == 0 || y1 == ty_n - 1;
                int c = border ? 1 : 0;
                for (int k2 = 0; k2 < 3 && !c; ++k2) {
                    const int ds = mvx_src_plane(d, Din, Dout, sd, pd, k2);
                    if (ds >= 0) c = hflag[(size_t)ds * ntiles + y0 * tx_n + x] | hflag[(size_t)ds * ntiles + y1 * tx_n + x];
                }
                r |= c;
            }
    }
    read[e] = r;
}

// ---- ... writing the reference's layout of the middle output instead: bev[frame][c * D + d][H][W] (modules/voxelnet/Pipe.py:40-41:
// the (C, D, H, W) result viewed as (C * D, H, W) -- a view there, a transposition of this library's channels-last storage).  The
// same values as bn_apply_tiles followed by cl_bev_transpose, without the channels-last tensor in between (288 MB written and
// read back per 4-frame step): the tile's 128 sites x C channels go through a padded LDS tile, every thread then stores four
// neighbouring sites of one channel (16 lanes = one 64-byte row piece of the tile).  C <= 64, one workgroup per (tile, plane).
__global__ __launch_bounds__(256) void bn_apply_tiles_bev(const float *__restrict__ y, const float *__restrict__ mi,
                                                          const float *__restrict__ c_bg, const int *__restrict__ tile_flags,
                                                          float *__restrict__ bev, int D, int H, int W, int C, int ntiles) {
    __shared__ float s_t[ATH * ATW][64 + 1];
    const int tiles_x = (W + ATW - 1) / ATW;
    const int t = blockIdx.x, d = blockIdx.y, frame = d / D, dd = d - frame * D;
    const int c4n = C >> 2, ct = threadIdx.x % c4n, st = threadIdx.x / c4n, spb = 256 / c4n;
    const int ty0 = (t / tiles_x) * ATH, tx0 = (t % tiles_x) * ATW;
    const bool on = tile_flags[(size_t)d * ntiles + t] != 0;          // block-uniform
    const float4 cb = *(const float4 *)(c_bg + (size_t)d * C + ct * 4);
    const float *fmi = mi + (size_t)frame * 2 * C;
    const float4 m = *(const float4 *)(fmi + ct * 4), iv = *(const float4 *)(fmi + C + ct * 4);
    for (int sidx = st; sidx < ATH * ATW; sidx += spb) {
        const int gy = ty0 + sidx / ATW, gx = tx0 + sidx % ATW;
        float4 o = cb;
        if (on && gy < H && gx < W) {
            const float4 v = *(const float4 *)(y + (((size_t)d * H + gy) * W + gx) * C + ct * 4);
            o = make_float4((v.x - m.x) * iv.x, (v.y - m.y) * iv.y, (v.z - m.z) * iv.z, (v.w - m.w) * iv.w);
        }
        s_t[sidx][ct * 4 + 0] = o.x; s_t[sidx][ct * 4 + 1] = o.y; s_t[sidx][ct * 4 + 2] = o.z; s_t[sidx][ct * 4 + 3] = o.w;
    }
    __syncthreads();
    // (tile row, channel, group of four columns): lanes walk the four groups of a 64-byte row piece, then the channels (LDS
    // banks: site + channel mod 32 with the 65-float pitch -- 28 distinct banks per wave)
    float *fb = bev + (size_t)frame * C * D * H * W;
    for (int e = threadIdx.x; e < C * ATH * (ATW / 4); e += 256) {
        const int g4 = e % (ATW / 4), c = (e / (ATW / 4)) % C, r = e / (C * (ATW / 4));
        const int gy = ty0 + r, gx = tx0 + g4 * 4;
        if (gy >= H || gx >= W) continue;
        const int s0 = r * ATW + g4 * 4;
        float *dst = fb + (((size_t)c * D + dd) * H + gy) * W + gx;
        if (gx + 3 < W && (W & 3) == 0) {
            *(float4 *)dst = make_float4(s_t[s0][c], s_t[s0 + 1][c], s_t[s0 + 2][c], s_t[s0 + 3][c]);
        } else {
            for (int j = 0; j < 4 && gx + j < W; ++j) dst[j] = s_t[s0 + j][c];
        }
    }
}

// ---- BatchNorm + ReLU backward restricted to the active tiles of a layer output -----------------------
// Outside the active tiles every site holds the background (y = y_bg[d], yhat = c[d]); its share of the
// batch sums follows from A[d][c] = sum over plane d of the incoming gradient (closed form, see
// mvx_conv3d_input_grad_sums) minus what the active tiles hold.  The gradient itself is only produced on
// the active tiles: nothing downstream reads it elsewhere (mvx_sparse_conv_gather_dz reads next to voxels).
constexpr int BREP = 8;

// tile_list[j] = d * ntiles + tile of the active tiles (ascending); n_act; n_inact[d] = sites of plane d in inactive tiles
constexpr int BNB_MAX_PLANES = 16 * MVX_MAX_FRAMES;
__global__ __launch_bounds__(1024) void bnb_tile_list(const int *__restrict__ tile_flags, int D, int H, int W, int ntiles,
                                                      int *__restrict__ list, int *__restrict__ n_act,
                                                      int *__restrict__ n_inact, unsigned *__restrict__ amax_slot) {
    __shared__ int smem[17];
    if (amax_slot && threadIdx.x == 0) *amax_slot = 0u;        // max |dz| of the apply pass starts from zero
    __shared__ int s_inact[BNB_MAX_PLANES];            // D here = ALL planes of the launch (frames x planes per frame)
    const int tiles_x = (W + ATW - 1) / ATW;
    if (threadIdx.x < BNB_MAX_PLANES) s_inact[threadIdx.x] = 0;
    __syncthreads();
    const int total = D * ntiles;
    int base = 0;
    // contiguous run per thread, flags in one 64-bit word, one block scan per 65,536 entries (see wgrad_step_list)
    for (int s0 = 0; s0 < total; s0 += 1024 * 64) {
        const int left = total - s0;
        const int per = left >= 1024 * 64 ? 64 : (left + 1023) / 1024;
        const int b0 = s0 + (int)threadIdx.x * per;
        unsigned long long bits = 0ull;
#pragma unroll 8
        for (int k = 0; k < per; ++k) {
            const int e = b0 + k;
            if (e < total) bits |= (unsigned long long)(tile_flags[e] != 0) << k;
        }
        for (int k = 0; k < per; ++k) {
            const int e = b0 + k;
            if (e < total && !((bits >> k) & 1ull)) {
                const int d = e / ntiles, t = e - d * ntiles;
                const int ty0 = (t / tiles_x) * ATH, tx0 = (t % tiles_x) * ATW;
                atomicAdd(&s_inact[d], min(ATH, H - ty0) * min(ATW, W - tx0));
            }
        }
        int tot;
        int pos = base + block_excl_scan_i32(__popcll(bits), smem, &tot);
        while (bits) {
            const int k = __ffsll((long long)bits) - 1;
            bits &= bits - 1;
            list[pos++] = b0 + k;
        }
        base += tot;
    }
    __syncthreads();
    if (threadIdx.x == 0) *n_act = base;
    if (threadIdx.x < D) n_inact[threadIdx.x] = s_inact[threadIdx.x];
}

// A workgroup takes a CONTIGUOUS run of the listed tiles (ascending (plane, tile) order: a run lies in one or two
// planes); thread = (site column, channel quad).  mode 0: sums[rep][0][d][c] += dyh, sums[rep][1][0][c] += dyh * (yhat -
// c_bg[d]).  mode 1: dz = (y > 0) * inv * (dyh - a - yhat * b), sums[rep][2][0][c] += dz.  The per-thread sums run over
// all tiles of the run and are reduced through LDS + f64 atomics only when the plane changes and at the end (the first
// form reduced and issued 128-192 atomics per TILE: 0.25 ms per launch at 1.4 TB/s algorithmic).
template <int MODE>
__global__ __launch_bounds__(256) void bnb_tiles(const float *__restrict__ dyh, const float *__restrict__ y,
                                                 const float *__restrict__ mi, const float *__restrict__ c_bg,
                                                 const float *__restrict__ ab, const int *__restrict__ list,
                                                 const int *__restrict__ n_act, int D, int H, int W, int C, int ntiles,
                                                 float *__restrict__ dz, double *__restrict__ sums_all,
                                                 unsigned *__restrict__ amax_slot) {
    __shared__ float red[2][256][4];
    float mx = 0.f;                                   // MODE 1: max |dz| this thread wrote (-> amax_slot, see mvx_wave_amax_to)
    const int tiles_x = (W + ATW - 1) / ATW;
    const int c4n = C >> 2, ct = threadIdx.x % c4n, st = threadIdx.x / c4n, spb = 256 / c4n;
    const int nact = *n_act;
    const int per = (nact + (int)gridDim.x - 1) / (int)gridDim.x;
    const int j0 = (int)blockIdx.x * per, j1 = min(nact, j0 + per);
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    float4 m = s1, iv = s1, cb = s1, a = s1, b = s1;
    int cur = -1;                                     // plane the running sums belong to
    auto flush = [&](int d) {                         // block-uniform
        const int frame = d / D, dl = d - frame * D;
        double *sums = sums_all + (size_t)frame * BREP * (D + 2) * C;
        __syncthreads();
        red[0][threadIdx.x][0] = s1.x; red[0][threadIdx.x][1] = s1.y; red[0][threadIdx.x][2] = s1.z; red[0][threadIdx.x][3] = s1.w;
        red[1][threadIdx.x][0] = s2.x; red[1][threadIdx.x][1] = s2.y; red[1][threadIdx.x][2] = s2.z; red[1][threadIdx.x][3] = s2.w;
        __syncthreads();
        if (st == 0) {
            const unsigned rep = blockIdx.x % BREP;
            double *base = sums + (size_t)rep * (D + 2) * C;        // layout per replica: [D planes of P1][Q1][Z1]
            for (int q4 = 0; q4 < 4; ++q4) {
                double t1 = 0.0, t2 = 0.0;
                for (int q = 0; q < spb; ++q) { t1 += (double)red[0][q * c4n + ct][q4]; t2 += (double)red[1][q * c4n + ct][q4]; }
                const int n = ct * 4 + q4;
                if (MODE == 0) {
                    atomicAdd(base + (size_t)dl * C + n, t1);
                    atomicAdd(base + (size_t)D * C + n, t2);
                } else {
                    atomicAdd(base + (size_t)(D + 1) * C + n, t1);
                }
            }
        }
        s1 = make_float4(0.f, 0.f, 0.f, 0.f); s2 = s1;
    };
    auto site = [&](const float4 g, const float4 v, size_t off) __attribute__((always_inline)) {
        const float4 yh = make_float4((v.x - m.x) * iv.x, (v.y - m.y) * iv.y, (v.z - m.z) * iv.z, (v.w - m.w) * iv.w);
        if (MODE == 0) {
            s1.x += g.x; s1.y += g.y; s1.z += g.z; s1.w += g.w;
            s2.x += g.x * (yh.x - cb.x); s2.y += g.y * (yh.y - cb.y); s2.z += g.z * (yh.z - cb.z); s2.w += g.w * (yh.w - cb.w);
        } else {
            float4 o;
            o.x = v.x > 0.f ? iv.x * (g.x - (a.x + yh.x * b.x)) : 0.f;
            o.y = v.y > 0.f ? iv.y * (g.y - (a.y + yh.y * b.y)) : 0.f;
            o.z = v.z > 0.f ? iv.z * (g.z - (a.z + yh.z * b.z)) : 0.f;
            o.w = v.w > 0.f ? iv.w * (g.w - (a.w + yh.w * b.w)) : 0.f;
            *(float4 *)(dz + off) = o;
            s1.x += o.x; s1.y += o.y; s1.z += o.z; s1.w += o.w;
            mx = fmaxf(fmaxf(mx, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
        }
    };
    for (int j = j0; j < j1; ++j) {
        // d = GLOBAL plane (frames stacked along depth, D planes each); per-frame: mean / inverse std, a / b, sums
        const int e = list[j], d = e / ntiles, t = e - d * ntiles;
        if (d != cur) {
            if (cur >= 0) flush(cur);
            cur = d;
            const int frame = d / D;
            const float *fmi = mi + (size_t)frame * 2 * C;
            m = *(const float4 *)(fmi + ct * 4); iv = *(const float4 *)(fmi + C + ct * 4);
            cb = *(const float4 *)(c_bg + (size_t)d * C + ct * 4);
            if (MODE == 1) { a = *(const float4 *)(ab + (size_t)frame * 2 * C + ct * 4); b = *(const float4 *)(ab + (size_t)frame * 2 * C + C + ct * 4); }
        }
        const int ty0 = (t / tiles_x) * ATH, tx0 = (t % tiles_x) * ATW;
        if (spb == ATW && ty0 + ATH <= H && tx0 + ATW <= W) {
            // full tile, 64 channels: thread = (column st, quad ct), the eight rows' loads in flight together
            float4 gq[ATH], vq[ATH];
            const size_t off0 = (((size_t)d * H + ty0) * W + tx0 + st) * C + ct * 4;
#pragma unroll
            for (int r = 0; r < ATH; ++r) {
                gq[r] = *(const float4 *)(dyh + off0 + (size_t)r * W * C);
                vq[r] = *(const float4 *)(y + off0 + (size_t)r * W * C);
            }
#pragma unroll
            for (int r = 0; r < ATH; ++r) site(gq[r], vq[r], off0 + (size_t)r * W * C);
        } else {
            for (int sidx = st; sidx < ATH * ATW; sidx += spb) {
                const int gy = ty0 + sidx / ATW, gx = tx0 + sidx % ATW;
                if (gy >= H || gx >= W) continue;
                const size_t off = (((size_t)d * H + gy) * W + gx) * C + ct * 4;
                site(*(const float4 *)(dyh + off), *(const float4 *)(y + off), off);
            }
        }
    }
    if (cur >= 0) flush(cur);
    if (MODE == 1 && amax_slot) mvx_wave_amax_to(amax_slot, mx);
}

// a = sum_d A[d] / N ; b = (Q1 + sum_d c[d] A[d]) / N
__global__ void bnb_finalize_ab(const double *__restrict__ sums, const float *__restrict__ A, const float *__restrict__ c_bg,
                                int D, int C, double count, float *__restrict__ ab) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= C) return;
    const int frame = blockIdx.y;                         // everything below is per frame
    sums += (size_t)frame * BREP * (D + 2) * C;
    A += (size_t)frame * D * C;
    c_bg += (size_t)frame * D * C;
    ab += (size_t)frame * 2 * C;
    double q1 = 0.0;
    for (int rep = 0; rep < BREP; ++rep) q1 += sums[((size_t)rep * (D + 2) + D) * C + n];
    double sa = 0.0, sca = 0.0;
    for (int d = 0; d < D; ++d) {
        const double Ad = (double)A[(size_t)d * C + n];
        sa += Ad;
        sca += (double)c_bg[(size_t)d * C + n] * Ad;
    }
    ab[n] = (float)(sa / count);
    ab[C + n] = (float)((q1 + sca) / count);
}

// dbias = Z1 + sum_d [y_bg[d] > 0] inv ((A[d] - P1[d]) - n_inact[d] (a + c[d] b))
__global__ void bnb_dbias(const double *__restrict__ sums, const float *__restrict__ A, const float *__restrict__ c_bg,
                          const float *__restrict__ y_bg, const float *__restrict__ mi, const float *__restrict__ ab,
                          const int *__restrict__ n_inact, int D, int C, float *__restrict__ dbias, int accumulate,
                          float *__restrict__ dz_inactive_sums, int n_frames) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= C) return;
    double t = 0.0;                                   // the bias gradient sums over the frames
    for (int frame = 0; frame < n_frames; ++frame) {
        const double *fs = sums + (size_t)frame * BREP * (D + 2) * C;
        const float *fmi = mi + (size_t)frame * 2 * C, *fab = ab + (size_t)frame * 2 * C;
        double z1 = 0.0;
        for (int rep = 0; rep < BREP; ++rep) z1 += fs[((size_t)rep * (D + 2) + D + 1) * C + n];
        const double inv = (double)fmi[C + n], a = (double)fab[n], b = (double)fab[C + n];
        t += z1;
        for (int dl = 0; dl < D; ++dl) {
            const int d = frame * D + dl;
            double s = 0.0;                              // sum of dz over the inactive tiles of plane d
            if (y_bg[(size_t)d * C + n] > 0.f) {
                double p1 = 0.0;
                for (int rep = 0; rep < BREP; ++rep) p1 += fs[((size_t)rep * (D + 2) + dl) * C + n];
                s = inv * (((double)A[(size_t)d * C + n] - p1) - (double)n_inact[d] * (a + (double)c_bg[(size_t)d * C + n] * b));
            }
            if (dz_inactive_sums) dz_inactive_sums[(size_t)d * C + n] = (float)s;
            t += s;
        }
    }
    if (dbias) dbias[n] = accumulate ? dbias[n] + (float)t : (float)t;
}

}  // namespace

extern "C" size_t mvx_bn_relu_backward_tiles_workspace_bytes_frames(int32_t planes, int32_t h, int32_t w, int32_t channels,
                                                                   int32_t n_frames) {
    if (planes <= 0 || planes > 16 || h <= 0 || w <= 0 || channels <= 0 || n_frames <= 0 || n_frames > MVX_MAX_FRAMES) return 0;
    const size_t ntiles = (size_t)mvx_cdiv(w, ATW) * mvx_cdiv(h, ATH);
    const size_t P = (size_t)planes * n_frames;
    return sizeof(double) * BREP * (planes + 2) * channels * n_frames + sizeof(float) * 2 * channels * n_frames +
           sizeof(int) * (P * ntiles + 16 + P);
}

extern "C" size_t mvx_bn_relu_backward_tiles_workspace_bytes(int32_t planes, int32_t h, int32_t w, int32_t channels) {
    return mvx_bn_relu_backward_tiles_workspace_bytes_frames(planes, h, w, channels, 1);
}

extern "C" int mvx_bn_relu_backward_tiles_frames(const float *dyhat, const float *y, const float *mean_inv, const float *c_bg,
                                                 const float *y_bg, const float *plane_grad_sums, const int32_t *tile_flags,
                                                 int32_t planes, int32_t h, int32_t w, int32_t channels, float *dz,
                                                 float *dbias, float *dz_inactive_sums, float *dz_amax, int32_t flags,
                                                 void *workspace, size_t workspace_bytes, int32_t n_frames, void *stream) {
    MVX_CHECK_ARG(dyhat && y && mean_inv && c_bg && y_bg && plane_grad_sums && tile_flags && dz && workspace);
    MVX_CHECK_ARG(planes > 0 && planes <= 16 && h > 0 && w > 0 && channels > 0 && channels % 4 == 0 && 256 % (channels / 4) == 0);
    MVX_CHECK_ARG(n_frames >= 1 && n_frames <= MVX_MAX_FRAMES);
    MVX_CHECK_ARG(workspace_bytes >= mvx_bn_relu_backward_tiles_workspace_bytes_frames(planes, h, w, channels, n_frames));
    hipStream_t st = (hipStream_t)stream;
    const int ntiles = (int)(mvx_cdiv(w, ATW) * mvx_cdiv(h, ATH));
    const int P = planes * n_frames;                   // all planes of the launch
    double *sums = (double *)workspace;
    float *ab = (float *)(sums + (size_t)BREP * (planes + 2) * channels * n_frames);
    int *list = (int *)(ab + 2 * (size_t)channels * n_frames);
    int *n_act = list + (size_t)P * ntiles, *n_inact = n_act + 16;
    hipError_t e = hipMemsetAsync(sums, 0, sizeof(double) * BREP * (planes + 2) * channels * n_frames, st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(bnb_tile_list, dim3(1), dim3(1024), 0, st, tile_flags, P, h, w, ntiles, list, n_act, n_inact, (unsigned *)dz_amax);
    MVX_LAUNCH_CHECK();
    const double count = (double)planes * h * w;       // per frame
    const unsigned grid = (unsigned)(P * ntiles > 2048 ? 2048 : P * ntiles);
    hipLaunchKernelGGL(bnb_tiles<0>, dim3(grid), dim3(256), 0, st, dyhat, y, mean_inv, c_bg, (const float *)ab, (const int *)list,
                       (const int *)n_act, planes, h, w, channels, ntiles, dz, sums, (unsigned *)nullptr);
    MVX_LAUNCH_CHECK();
    hipLaunchKernelGGL(bnb_finalize_ab, dim3(mvx_cdiv(channels, 64), n_frames), dim3(64), 0, st, (const double *)sums,
                       plane_grad_sums, c_bg, planes, channels, count, ab);
    MVX_LAUNCH_CHECK();
    hipLaunchKernelGGL(bnb_tiles<1>, dim3(grid), dim3(256), 0, st, dyhat, y, mean_inv, c_bg, (const float *)ab, (const int *)list,
                       (const int *)n_act, planes, h, w, channels, ntiles, dz, sums, (unsigned *)dz_amax);
    MVX_LAUNCH_CHECK();
    if (dbias || dz_inactive_sums) {
        hipLaunchKernelGGL(bnb_dbias, dim3(mvx_cdiv(channels, 64)), dim3(64), 0, st, (const double *)sums, plane_grad_sums, c_bg,
                           y_bg, mean_inv, (const float *)ab, (const int *)n_inact, planes, channels, dbias,
                           flags & MVX_FLAG_ACCUMULATE, dz_inactive_sums, n_frames);
        MVX_LAUNCH_CHECK();
    }
    return MVX_OK;
}

extern "C" int mvx_bn_relu_backward_tiles(const float *dyhat, const float *y, const float *mean_inv, const float *c_bg,
                                          const float *y_bg, const float *plane_grad_sums, const int32_t *tile_flags,
                                          int32_t planes, int32_t h, int32_t w, int32_t channels, float *dz, float *dbias,
                                          float *dz_inactive_sums, int32_t flags, void *workspace,
                                          size_t workspace_bytes, void *stream) {
    return mvx_bn_relu_backward_tiles_frames(dyhat, y, mean_inv, c_bg, y_bg, plane_grad_sums, tile_flags, planes, h, w, channels,
                                             dz, dbias, dz_inactive_sums, nullptr, flags, workspace, workspace_bytes, 1, stream);
}

extern "C" int mvx_activity_dilate_frames(const void *src, int32_t src_is_index, int32_t din, int32_t dout, int32_t h, int32_t w,
                                          int32_t stride_d, int32_t pad_d, int32_t mark_border, uint8_t *dst_mask,
                                          int32_t *dst_halo_flags, int32_t *dst_tile_flags, int32_t n_frames, void *stream) {
    MVX_CHECK_ARG(src && dst_mask && din > 0 && dout > 0 && h > 0 && w > 0);
    MVX_CHECK_ARG(stride_d >= 1 && stride_d <= 2 && pad_d >= 0 && pad_d <= 1);
    MVX_CHECK_ARG(dout == (din + 2 * pad_d - 3) / stride_d + 1);
    MVX_CHECK_ARG(n_frames >= 1 && n_frames <= MVX_MAX_FRAMES);
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)n_frames * dout * h * w;
    if (w % 4 == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst_mask & 3) == 0)
        hipLaunchKernelGGL(activity_sites4, dim3(mvx_cdiv(n / 4, 256) > 4096 ? 4096 : mvx_cdiv(n / 4, 256)), dim3(256), 0, st, src,
                           src_is_index, din, dout, h, w, stride_d, pad_d, mark_border, dst_mask, n_frames);
    else
        hipLaunchKernelGGL(activity_sites, dim3(mvx_cdiv(n, 256) > 4096 ? 4096 : mvx_cdiv(n, 256)), dim3(256), 0, st, src,
                           src_is_index, din, dout, h, w, stride_d, pad_d, mark_border, dst_mask, n_frames);
    MVX_LAUNCH_CHECK();
    if (dst_halo_flags || dst_tile_flags) {
        hipLaunchKernelGGL(activity_halo_flags, dim3(mvx_cdiv(w, ATW) * mvx_cdiv(h, ATH), dout * n_frames), dim3(256), 0, st,
                           (const unsigned char *)dst_mask, dout * n_frames, h, w, dst_halo_flags, dst_tile_flags);
        MVX_LAUNCH_CHECK();
    }
    return MVX_OK;
}

extern "C" int mvx_activity_dilate(const void *src, int32_t src_is_index, int32_t din, int32_t dout, int32_t h, int32_t w,
                                   int32_t stride_d, int32_t pad_d, int32_t mark_border, uint8_t *dst_mask,
                                   int32_t *dst_halo_flags, int32_t *dst_tile_flags, void *stream) {
    return mvx_activity_dilate_frames(src, src_is_index, din, dout, h, w, stride_d, pad_d, mark_border, dst_mask,
                                      dst_halo_flags, dst_tile_flags, 1, stream);
}

extern "C" int mvx_tile_dilate_flags_frames(const int32_t *in_tile_flags, const int32_t *self_tile_flags, int32_t din,
                                            int32_t dout, int32_t h, int32_t w, int32_t stride_d, int32_t pad_d,
                                            int32_t *out_tile_flags, int32_t n_frames, void *stream) {
    MVX_CHECK_ARG(in_tile_flags && out_tile_flags && din > 0 && dout > 0 && h > 0 && w > 0);
    MVX_CHECK_ARG(n_frames >= 1 && n_frames <= MVX_MAX_FRAMES);
    const int ty = (int)mvx_cdiv(h, ATH), tx = (int)mvx_cdiv(w, ATW);
    hipLaunchKernelGGL(tile_dilate_flags, dim3(mvx_cdiv((long long)n_frames * dout * ty * tx, 256)), dim3(256), 0,
                       (hipStream_t)stream, in_tile_flags, self_tile_flags, din, dout, ty, tx, stride_d, pad_d, out_tile_flags,
                       n_frames);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" int mvx_tile_dilate_flags(const int32_t *in_tile_flags, const int32_t *self_tile_flags, int32_t din, int32_t dout,
                                     int32_t h, int32_t w, int32_t stride_d, int32_t pad_d, int32_t *out_tile_flags,
                                     void *stream) {
    return mvx_tile_dilate_flags_frames(in_tile_flags, self_tile_flags, din, dout, h, w, stride_d, pad_d, out_tile_flags, 1,
                                        stream);
}

extern "C" int mvx_conv3d_background_frames(const float *w, const float *c_in, int32_t din, int32_t dout, int32_t cin,
                                            int32_t cout, int32_t stride_d, int32_t pad_d, float *bg_pre, int32_t n_frames,
                                            void *stream) {
    MVX_CHECK_ARG(w && c_in && bg_pre && din > 0 && dout > 0 && cin > 0 && cout > 0);
    MVX_CHECK_ARG(n_frames >= 1 && n_frames <= MVX_MAX_FRAMES);
    hipLaunchKernelGGL(conv_background, dim3(cout, dout * n_frames), dim3(64), 0, (hipStream_t)stream, w, c_in, din, dout, cin,
                       cout, stride_d, pad_d, bg_pre, (float *)nullptr, (float *)nullptr);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

// bg f32: [planes][cout] totals | [planes][3][cout] per-depth-tap constants | [planes][9][cout] image-border classes
// (planes = dout * n_frames; 13 * planes * cout floats): the buffer mvx_conv3d_forward_bg_frames takes with MVX_FLAG_BG_TAPS
extern "C" int mvx_conv3d_background_taps_frames(const float *w, const float *c_in, int32_t din, int32_t dout, int32_t cin,
                                                 int32_t cout, int32_t stride_d, int32_t pad_d, float *bg, int32_t n_frames,
                                                 void *stream) {
    MVX_CHECK_ARG(w && c_in && bg && din > 0 && dout > 0 && cin > 0 && cout > 0);
    MVX_CHECK_ARG(n_frames >= 1 && n_frames <= MVX_MAX_FRAMES);
    hipLaunchKernelGGL(conv_background, dim3(cout, dout * n_frames), dim3(64), 0, (hipStream_t)stream, w, c_in, din, dout, cin,
                       cout, stride_d, pad_d, bg, bg + (size_t)dout * n_frames * cout, bg + (size_t)4 * dout * n_frames * cout);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" int mvx_conv3d_background(const float *w, const float *c_in, int32_t din, int32_t dout, int32_t cin,
                                     int32_t cout, int32_t stride_d, int32_t pad_d, float *bg_pre, void *stream) {
    return mvx_conv3d_background_frames(w, c_in, din, dout, cin, cout, stride_d, pad_d, bg_pre, 1, stream);
}

extern "C" int mvx_bn_background_frames(const float *bg_pre, const float *bias, const float *mean_inv, int32_t planes,
                                        int32_t channels, int32_t flags, float *y_bg, float *c_out, int32_t n_frames,
                                        void *stream) {
    MVX_CHECK_ARG(mean_inv && c_out && planes > 0 && channels > 0 && n_frames >= 1 && n_frames <= MVX_MAX_FRAMES);
    hipLaunchKernelGGL(bn_background, dim3(mvx_cdiv((long long)n_frames * planes * channels, 256)), dim3(256), 0,
                       (hipStream_t)stream, bg_pre, bias, mean_inv, planes, channels, flags & MVX_FLAG_RELU, y_bg, c_out, n_frames);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" int mvx_bn_apply_tiles_frames(const float *y, const float *mean_inv, const float *c_bg, const int32_t *tile_flags,
                                         float *out, int32_t planes, int32_t h, int32_t w, int32_t channels, int32_t n_frames,
                                         void *stream) {
    MVX_CHECK_ARG(y && mean_inv && c_bg && tile_flags && out && planes > 0 && h > 0 && w > 0);
    MVX_CHECK_ARG(channels > 0 && channels % 4 == 0 && channels <= 1024 && 256 % (channels / 4) == 0);
    MVX_CHECK_ARG(n_frames >= 1 && n_frames <= MVX_MAX_FRAMES);
    const int ntiles = (int)(mvx_cdiv(w, ATW) * mvx_cdiv(h, ATH));
    hipLaunchKernelGGL(bn_apply_tiles, dim3(ntiles, planes * n_frames), dim3(256), 0, (hipStream_t)stream, y, mean_inv, c_bg,
                       (const int *)tile_flags, out, planes, h, w, channels, ntiles, (const int *)nullptr);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

// ... that leaves the background tiles no consumer reads unwritten: read_flags [n_frames * planes][tiles] from mvx_tile_read_flags_frames
extern "C" int mvx_bn_apply_tiles_read_frames(const float *y, const float *mean_inv, const float *c_bg, const int32_t *tile_flags,
                                              const int32_t *read_flags, float *out, int32_t planes, int32_t h, int32_t w,
                                              int32_t channels, int32_t n_frames, void *stream) {
    MVX_CHECK_ARG(y && mean_inv && c_bg && tile_flags && read_flags && out && planes > 0 && h > 0 && w > 0);
    MVX_CHECK_ARG(channels > 0 && channels % 4 == 0 && channels <= 1024 && 256 % (channels / 4) == 0);
    MVX_CHECK_ARG(n_frames >= 1 && n_frames <= MVX_MAX_FRAMES);
    const int ntiles = (int)(mvx_cdiv(w, ATW) * mvx_cdiv(h, ATH));
    hipLaunchKernelGGL(bn_apply_tiles, dim3(ntiles, planes * n_frames), dim3(256), 0, (hipStream_t)stream, y, mean_inv, c_bg,
                       (const int *)tile_flags, out, planes, h, w, channels, ntiles, (const int *)read_flags);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

// read_flags [n_frames * din][tiles] of a layer's INPUT from the halo flags of that input (mvx_activity_dilate_frames) and the layer's
// depth geometry: non-zero = the layer's background-aware forward / weight gradient may read the tile (a superset)
extern "C" int mvx_tile_read_flags_frames(const int32_t *in_halo_flags, int32_t din, int32_t dout, int32_t h, int32_t w,
                                          int32_t stride_d, int32_t pad_d, int32_t *read_flags, int32_t n_frames, void *stream) {
    MVX_CHECK_ARG(in_halo_flags && read_flags && din > 0 && dout > 0 && h > 0 && w > 0);
    MVX_CHECK_ARG(n_frames >= 1 && n_frames <= MVX_MAX_FRAMES && stride_d >= 1 && stride_d <= 2 && pad_d >= 0 && pad_d <= 1);
    const int ty = (int)mvx_cdiv(h, ATH), tx = (int)mvx_cdiv(w, ATW);
    hipLaunchKernelGGL(tile_read_flags, dim3(mvx_cdiv((long long)n_frames * din * ty * tx, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const int *)in_halo_flags, din, dout, ty, tx, stride_d, pad_d, (int *)read_flags, n_frames);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" int mvx_bn_apply_tiles_bev_frames(const float *y, const float *mean_inv, const float *c_bg, const int32_t *tile_flags,
                                             float *bev, int32_t planes, int32_t h, int32_t w, int32_t channels, int32_t n_frames,
                                             void *stream) {
    MVX_CHECK_ARG(y && mean_inv && c_bg && tile_flags && bev && planes > 0 && h > 0 && w > 0);
    MVX_CHECK_ARG(channels > 0 && channels % 4 == 0 && channels <= 64 && 256 % (channels / 4) == 0);
    MVX_CHECK_ARG(n_frames >= 1 && n_frames <= MVX_MAX_FRAMES && (((uintptr_t)bev) & 15) == 0);
    const int ntiles = (int)(mvx_cdiv(w, ATW) * mvx_cdiv(h, ATH));
    hipLaunchKernelGGL(bn_apply_tiles_bev, dim3(ntiles, planes * n_frames), dim3(256), 0, (hipStream_t)stream, y, mean_inv, c_bg,
                       (const int *)tile_flags, bev, planes, h, w, channels, ntiles);
    MVX_LAUNCH_CHECK();
    return MVX_OK;
}

extern "C" int mvx_bn_background(const float *bg_pre, const float *bias, const float *mean_inv, int32_t planes,
                                 int32_t channels, int32_t flags, float *y_bg, float *c_out, void *stream) {
    return mvx_bn_background_frames(bg_pre, bias, mean_inv, planes, channels, flags, y_bg, c_out, 1, stream);
}
