// mfma_dep.hip - cost of DEPENDENT v_mfma_f32_32x32x16_f16 chains (every MFMA accumulates into the result of the previous one) against
// 2 / 4 interleaved accumulators, with 0 or 6 VALU fillers per MFMA, 1 / 2 / 3 waves per SIMD.  gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define M(acc) "v_mfma_f32_32x32x16_f16 %[" #acc "], %[a], %[b], %[" #acc "]\n"
#define F6 "v_fma_f32 %[f0], %[f0], %[m], %[c]\n v_fma_f32 %[f1], %[f1], %[m], %[c]\n v_fma_f32 %[f2], %[f2], %[m], %[c]\n v_fma_f32 %[f3], %[f3], %[m], %[c]\n v_fma_f32 %[f0], %[f0], %[m], %[c]\n v_fma_f32 %[f1], %[f1], %[m], %[c]\n"
#define F0
#define KERNEL(name, BODY)                                                                                         \
    __global__ __launch_bounds__(768) void name(int iters, float* out) {                                             \
        f32x16 x0 = {}, x1 = {}, x2 = {}, x3 = {};                                                                   \
        f16x8 a, b;                                                                                                  \
        for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 1e-3f + j); b[j] = (_Float16)1; }             \
        float f0 = threadIdx.x, f1 = 1.f, f2 = 2.f, f3 = 3.f, m = 1.0000001f, c = 1e-9f;                            \
        for (int i = 0; i < iters; ++i)                                                                              \
            asm volatile(BODY : [x0] "+v"(x0), [x1] "+v"(x1), [x2] "+v"(x2), [x3] "+v"(x3), [f0] "+v"(f0), [f1] "+v"(f1), [f2] "+v"(f2), [f3] "+v"(f3) \
                         : [a] "v"(a), [b] "v"(b), [m] "v"(m), [c] "v"(c));                                          \
        asm volatile("s_nop 15\n s_nop 15\n" ::: "memory");                                                         \
        float r = x0[0] + x1[1] + x2[2] + x3[3] + f0 + f1 + f2 + f3;                                                 \
        if (r == 123.456f) out[threadIdx.x] = r;                                                                     \
    }
KERNEL(dep1_f0, M(x0) F0 M(x0) F0 M(x0) F0 M(x0) F0 M(x0) F0 M(x0) F0 M(x0) F0 M(x0) F0)
KERNEL(dep2_f0, M(x0) F0 M(x1) F0 M(x0) F0 M(x1) F0 M(x0) F0 M(x1) F0 M(x0) F0 M(x1) F0)
KERNEL(dep4_f0, M(x0) F0 M(x1) F0 M(x2) F0 M(x3) F0 M(x0) F0 M(x1) F0 M(x2) F0 M(x3) F0)
KERNEL(blk2_f0, M(x0) F0 M(x0) F0 M(x0) F0 M(x0) F0 M(x1) F0 M(x1) F0 M(x1) F0 M(x1) F0)
KERNEL(dep1_f6, M(x0) F6 M(x0) F6 M(x0) F6 M(x0) F6 M(x0) F6 M(x0) F6 M(x0) F6 M(x0) F6)
KERNEL(dep2_f6, M(x0) F6 M(x1) F6 M(x0) F6 M(x1) F6 M(x0) F6 M(x1) F6 M(x0) F6 M(x1) F6)
KERNEL(dep4_f6, M(x0) F6 M(x1) F6 M(x2) F6 M(x3) F6 M(x0) F6 M(x1) F6 M(x2) F6 M(x3) F6)
KERNEL(blk2_f6, M(x0) F6 M(x0) F6 M(x0) F6 M(x0) F6 M(x1) F6 M(x1) F6 M(x1) F6 M(x1) F6)
typedef void (*kern_t)(int, float*);
static void run(const char* name, kern_t k, float* d) {
    const int iters = 4000;
    printf("%-10s", name);
    for (int wps = 1; wps <= 3; ++wps) {
        hipLaunchKernelGGL(k, dim3(256), dim3(256 * wps), 0, 0, 10, d);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(256 * wps), 0, 0, iters, d);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("  %dw/SIMD %6.1f cyc/MFMA/SIMD", wps, ms * 1e-3 * 2.4e9 / (iters * 8.0 * wps));
    }
    printf("\n");
}
int main() {
    float* d; (void)hipMalloc(&d, 1 << 16);
    run("dep1_f0", dep1_f0, d); run("dep2_f0", dep2_f0, d); run("dep4_f0", dep4_f0, d); run("blk2_f0", blk2_f0, d);
    run("dep1_f6", dep1_f6, d); run("dep2_f6", dep2_f6, d); run("dep4_f6", dep4_f6, d); run("blk2_f6", blk2_f6, d);
    return 0;
}
