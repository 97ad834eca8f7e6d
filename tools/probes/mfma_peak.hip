// Register-only MFMA loop: the matrix-core rate this chip actually sustains (clock under load
// included), as the yardstick for the conv kernels' roofline fractions.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_peak.hip -o tools/probes/mfma_peak.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NC>
__global__ __launch_bounds__(256) void k_f32(float *out, int iters, float a, float b) {
    if (b < 0.f) {      // random-looking per-lane operands: data-dependent power draw
        unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
        a = (float)(h >> 8) * (1.0f / 16777216.0f) - 0.5f;
        h = h * 2246822519u + 12345u;
        b = (float)(h >> 8) * (1.0f / 16777216.0f) - 0.5f;
    }
    f32x16 acc[NC];
    for (int j = 0; j < NC; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 32 / NC; ++u)
#pragma unroll
            for (int j = 0; j < NC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < NC; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// the other f32 shape: 16x16x4 (8 passes = 32 cycles, 2,048 FLOP: the same 64 FLOP/clk/SIMD), NC chains of float4 accumulators
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <int NC>
__global__ __launch_bounds__(256) void k_f32_16(float *out, int iters, float a, float b) {
    if (b < 0.f) {
        unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
        a = (float)(h >> 8) * (1.0f / 16777216.0f) - 0.5f;
        h = h * 2246822519u + 12345u;
        b = (float)(h >> 8) * (1.0f / 16777216.0f) - 0.5f;
    }
    f32x4v acc[NC];
    for (int j = 0; j < NC; ++j) for (int r = 0; r < 4; ++r) acc[j][r] = 0.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 64 / NC; ++u)
#pragma unroll
            for (int j = 0; j < NC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < NC; ++j) for (int r = 0; r < 4; ++r) s += acc[j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_bf16(float *out, int iters, float a) {
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    bf16x8 x, y;
    for (int r = 0; r < 8; ++r) { x[r] = (__bf16)a; y[r] = (__bf16)(a + r); }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float *out; hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves = 1; waves <= 2; ++waves) {
        const int grid = 256 * waves;              // 4 waves per WG -> `waves` per SIMD
        for (int rep = 0; rep < 3; ++rep) {
            const int iters = 20000;
            float ms; double fl = (double)grid * 4 * iters * 32 * (2.0 * 32 * 32 * 2);
            hipEventRecord(e0); k_f32<4><<<grid, 256>>>(out, iters, 1.f, 2.f); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            printf("f32 32x32x2 4 chains  waves/SIMD=%d  %.2f ms  %.1f TFLOP/s\n", waves, ms, fl / ms / 1e9);
            hipEventRecord(e0); k_f32<2><<<grid, 256>>>(out, iters, 1.f, 2.f); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            printf("f32 32x32x2 2 chains  waves/SIMD=%d  %.2f ms  %.1f TFLOP/s\n", waves, ms, fl / ms / 1e9);
            hipEventRecord(e0); k_f32<4><<<grid, 256>>>(out, iters, 1.f, -2.f); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            printf("f32 32x32x2 4 chains RANDOM operands waves/SIMD=%d  %.2f ms  %.1f TFLOP/s\n", waves, ms, fl / ms / 1e9);
            hipEventRecord(e0); k_f32<1><<<grid, 256>>>(out, iters, 1.f, 2.f); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            printf("f32 32x32x2 1 chain   waves/SIMD=%d  %.2f ms  %.1f TFLOP/s\n", waves, ms, fl / ms / 1e9);
            hipEventRecord(e0); k_f32_16<8><<<grid, 256>>>(out, iters, 1.f, -2.f); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            {
                const double fl16 = (double)grid * 4 * iters * 64 * (2.0 * 16 * 16 * 4);
                printf("f32 16x16x4 8 chains RANDOM operands waves/SIMD=%d  %.2f ms  %.1f TFLOP/s\n", waves, ms, fl16 / ms / 1e9);
            }
            hipEventRecord(e0); k_bf16<<<grid, 256>>>(out, iters, 1.f); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            fl = (double)grid * 4 * iters * 32 * (2.0 * 32 * 32 * 16);
            printf("bf16 32x32x16 waves/SIMD=%d  %.2f ms  %.1f TFLOP/s\n", waves, ms, fl / ms / 1e9);
        }
    }
    return 0;
}
