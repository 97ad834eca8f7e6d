// Split arithmetic on the bf16 matrix cores, shared by conv3d_split.hip and linear_split.hip.
//
// An f32 operand is cut into NP bf16 pieces while it is staged (round to nearest even; every remainder is exact in f32):
//   NP = 2 ("bf16x3"): x ~ hi + lo (16 mantissa bits); a product is hi*hi + hi*lo + lo*hi: three v_mfma_f32_32x32x16_bf16,
//                      ~2e-5 relative error per product;
//   NP = 3 ("bf16x6"): x = hi + mid + lo EXACTLY (3 x 8 bits = the f32 mantissa); a product is hh + hm + mh + hl + lh + mm:
//                      six MFMAs; the dropped terms (ml, lm, ll) are below 2^-25 |x w|, under the rounding of an f32
//                      product -- fp32-grade accuracy at 6/16 of the exact-f32 MFMA's matrix cycles.
// Every bf16 x bf16 product is exact in the f32 accumulator; the small terms are accumulated first.
#pragma once
#include <hip/hip_runtime.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));      // also the raw container of eight 16-bit pieces of either format
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// FMT: the 16-bit format of the pieces.  0 = bf16 (8 mantissa bits per piece: NP = 2 "bf16x3", NP = 3 "bf16x6").
// 1 = fp16 (11 bits per piece): NP = 2 gives 22 mantissa bits -- x = hi + lo to 2^-22 -- and hh + hl + lh drops only the
// 2^-22 term: fp32-grade products from THREE MFMAs ("fp16x3"), provided the operands sit inside fp16's range (max 65,504;
// below 6.1e-5 a piece is subnormal and keeps an ABSOLUTE precision of 2^-25): the callers scale by powers of two.

// ---- operand scaling of the fp16 format (FMT = 1).  fp16 has 5 exponent bits: a piece pair covers an operand to 2^-22 only
// while its low piece stays normal, i.e. for |x| in [2^-3, 65504].  Every scale below is a POWER OF TWO, so scaling an operand
// and un-scaling the f32 accumulator changes no rounding: the result is the same function of the inputs, just computed where
// fp16 has its bits.  Three kinds of operand:
//   - gradients (dz) and externally supplied features: the caller binds the address of the tensor's max |value| (one float on
//     the device, written by the kernel that produced the tensor: mvx_split_operand_amax in include/mvx_hip.h) and the kernel
//     scales by 2^(14 - floor(log2 amax)): amax lands in [2^14, 2^15) -- the top of fp16's range, because amax is exact (the
//     largest value actually written) and gradient tensors have outlier rows (the shared padded row of the fusion MLP stands
//     for ~6e5 dense rows and its dz is that much larger than a real row's): elements down to amax / 2^17 keep 22 bits, one
//     bit less per binade below that;
//   - weights: the fixed factor SPLIT_F16_WSCALE = 2^8 (|w| < 255; 22 bits down to |w| = 2^-11);
//   - BatchNorm outputs and anything else without a bound amax: as is (|x| < 65504 is the caller's side of the contract).
// (struct SplitAmax { const float *a, *b; } -- first / second f32 operand of the launch, or NULL -- is declared in common.h)
constexpr float SPLIT_F16_WSCALE = 256.f;

__device__ __forceinline__ float split_scale_of(const float *amax) {
    if (!amax) return 1.f;
    const int e = (int)((__float_as_uint(*amax) >> 23) & 0xffu);
    if (e == 0 || e == 255) return 1.f;                      // zero / subnormal / inf / nan: not scaled
    int se = 268 - e;                                        // biased exponent of 2^(14 - (e - 127)): amax lands in [2^14, 2^15)
    se = se < 2 ? 2 : (se > 252 ? 252 : se);
    return __uint_as_float((unsigned)se << 23);
}
// The COARSE scale of a forward operand that comes from outside the library (sampled image features): the exponent is rounded
// down to a multiple of 8 binades, amax lands in [2^7, 2^15).  A forward scale must not depend on which tensor the executor
// happens to hold -- a frame set and one of its frames have different maxima, different fine scales would round elements with
// subnormal low pieces differently and flip ReLUs between the two executors -- and with 8-binade steps it does not, unless
// two maxima straddle a step.  Elements down to amax / 2^10 keep 22 bits in the worst case.
__device__ __forceinline__ float split_scale_coarse(const float *amax) {
    if (!amax) return 1.f;
    const int e = (int)((__float_as_uint(*amax) >> 23) & 0xffu);
    if (e == 0 || e == 255) return 1.f;
    const int t = 141 - e;                                   // 14 - floor(log2 amax)
    const int k = t >= 0 ? (t & ~7) : -((7 - t) & ~7);       // rounded down to a multiple of 8
    int se = 127 + k;
    se = se < 2 ? 2 : (se > 252 ? 252 : se);
    return __uint_as_float((unsigned)se << 23);
}
// 1 / s for s = 2^k (exact)
__device__ __forceinline__ float split_inverse(float s) { return __uint_as_float((254u << 23) - __float_as_uint(s)); }

template <int NP, int FMT = 0>
__device__ __forceinline__ void split_n(float x0, float x1, float x2, float x3, uint2 (&out)[NP]) {
    float r[4] = {x0, x1, x2, x3};
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        unsigned short p[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (FMT == 0) {
                const __bf16 b = (__bf16)r[j];
                p[j] = __builtin_bit_cast(unsigned short, b);
                r[j] -= (float)b;
            } else {
                const _Float16 b = (_Float16)r[j];
                p[j] = __builtin_bit_cast(unsigned short, b);
                r[j] -= (float)b;
            }
        }
        out[q].x = (unsigned)p[0] | ((unsigned)p[1] << 16);
        out[q].y = (unsigned)p[2] | ((unsigned)p[3] << 16);
    }
}

template <int FMT>
__device__ __forceinline__ f32x16 mvx_mfma16(const bf16x8 a, const bf16x8 b, const f32x16 c) {
    if constexpr (FMT == 0) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
#define MVX_SPLIT_MFMA(acc, x, y) acc = mvx_mfma16<FMT>(x, y, acc)

// acc += a * b
template <int NP, int FMT = 0>
__device__ __forceinline__ void split_mac1(f32x16 &acc, const bf16x8 (&a)[NP], const bf16x8 (&b)[NP]) {
    if constexpr (NP == 2) {
        MVX_SPLIT_MFMA(acc, a[1], b[0]); MVX_SPLIT_MFMA(acc, a[0], b[1]); MVX_SPLIT_MFMA(acc, a[0], b[0]);
    } else {
        MVX_SPLIT_MFMA(acc, a[0], b[2]); MVX_SPLIT_MFMA(acc, a[2], b[0]); MVX_SPLIT_MFMA(acc, a[1], b[1]);
        MVX_SPLIT_MFMA(acc, a[0], b[1]); MVX_SPLIT_MFMA(acc, a[1], b[0]); MVX_SPLIT_MFMA(acc, a[0], b[0]);
    }
}

// acc0 += a * b0, acc1 += a * b1 (two output tiles share the A fragments; the two chains are interleaved)
template <int NP, int FMT = 0>
__device__ __forceinline__ void split_mac2(f32x16 &acc0, f32x16 &acc1, const bf16x8 (&a)[NP], const bf16x8 (&b0)[NP],
                                           const bf16x8 (&b1)[NP]) {
#define MVX_T(i, j) MVX_SPLIT_MFMA(acc0, a[i], b0[j]); MVX_SPLIT_MFMA(acc1, a[i], b1[j]);
    if constexpr (NP == 2) {
        MVX_T(1, 0) MVX_T(0, 1) MVX_T(0, 0)
    } else {
        MVX_T(0, 2) MVX_T(2, 0) MVX_T(1, 1) MVX_T(0, 1) MVX_T(1, 0) MVX_T(0, 0)
    }
#undef MVX_T
}
