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
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NP>
__device__ __forceinline__ void split_n(float x0, float x1, float x2, float x3, uint2 (&out)[NP]) {
    float r[4] = {x0, x1, x2, x3};
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        unsigned short p[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const __bf16 b = (__bf16)r[j];
            p[j] = __builtin_bit_cast(unsigned short, b);
            r[j] -= (float)b;
        }
        out[q].x = (unsigned)p[0] | ((unsigned)p[1] << 16);
        out[q].y = (unsigned)p[2] | ((unsigned)p[3] << 16);
    }
}

#define MVX_SPLIT_MFMA(acc, x, y) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc, 0, 0, 0)

// acc += a * b
template <int NP>
__device__ __forceinline__ void split_mac1(f32x16 &acc, const bf16x8 (&a)[NP], const bf16x8 (&b)[NP]) {
    if constexpr (NP == 2) {
        MVX_SPLIT_MFMA(acc, a[1], b[0]); MVX_SPLIT_MFMA(acc, a[0], b[1]); MVX_SPLIT_MFMA(acc, a[0], b[0]);
    } else {
        MVX_SPLIT_MFMA(acc, a[0], b[2]); MVX_SPLIT_MFMA(acc, a[2], b[0]); MVX_SPLIT_MFMA(acc, a[1], b[1]);
        MVX_SPLIT_MFMA(acc, a[0], b[1]); MVX_SPLIT_MFMA(acc, a[1], b[0]); MVX_SPLIT_MFMA(acc, a[0], b[0]);
    }
}

// acc0 += a * b0, acc1 += a * b1 (two output tiles share the A fragments; the two chains are interleaved)
template <int NP>
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
