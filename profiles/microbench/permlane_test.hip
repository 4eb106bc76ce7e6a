// permlane_test.hip - cross-row sums on the VALU: x[l] + x[l^16] (+ x[l^32] + x[l^48]) via v_permlane16/32_swap (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ float sum_xor16(float x) {
    float a = x, b;
    asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "=&v"(b));
    return a + b;
}
__device__ __forceinline__ float sum_xor32(float x) {
    float a = x, b;
    asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "=&v"(b));
    return a + b;
}
__global__ void k(float* o) {
    float x = (float)(threadIdx.x * threadIdx.x);
    o[threadIdx.x] = sum_xor16(x);
    o[64 + threadIdx.x] = sum_xor32(sum_xor16(x));
}
int main() {
    float* d; (void)hipMalloc(&d, 512); k<<<1, 64>>>(d); float h[128]; (void)hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        float e16 = (float)(l * l + (l ^ 16) * (l ^ 16));
        float e64 = 0; for (int m = 0; m < 4; ++m) { int j = (l & 15) + 16 * m; e64 += (float)(j * j); }
        if (h[l] != e16 || h[64 + l] != e64) { if (bad < 4) printf("lane %d: got %g %g expected %g %g\n", l, h[l], h[64 + l], e16, e64); ++bad; }
    }
    printf("permlane swaps: %s\n", bad ? "MISMATCH" : "OK");
    return bad != 0;
}
