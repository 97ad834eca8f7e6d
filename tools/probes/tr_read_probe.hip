// Probe of ds_read_b64_tr_b16 lane mapping on gfx950 (developer tool).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(int* o) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = i;     // value = row*32 + col (pitch 32 elements)
    __syncthreads();
    const int lane = threadIdx.x;
    const int i16 = lane & 15, q = i16 >> 2, p = i16 & 3, g = lane >> 4;
    __attribute__((address_space(3))) s16x4* ptr =
        (__attribute__((address_space(3))) s16x4*)(lds + (g * 4 + q) * 32 + 4 * p);   // group g reads rows 4g..4g+3, cols 0..15
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(ptr);
    for (int j = 0; j < 4; ++j) o[lane * 4 + j] = v[j];
}
int main() {
    int* d; hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    int h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 1) {
        printf("lane %2d:", l);
        for (int j = 0; j < 4; ++j) printf(" (r%d,c%d)", h[l * 4 + j] / 32, h[l * 4 + j] % 32);
        printf("\n");
    }
    return 0;
}
