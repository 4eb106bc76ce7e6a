// coexec.hip - does a VALU-only wave overlap with an MFMA-only wave on the same SIMD?  (gfx950)
// Each workgroup = 8 waves on one CU (2 per SIMD): waves 0-3 run MFMA chains, waves 4-7 run v_fma chains.
// Modes: MFMA only, VALU only, both. If the pipes are independent, "both" takes max(t_mfma, t_valu); if they share
// execution resources it takes the sum.   build: hipcc --offload-arch=gfx950 -O3 coexec.hip -o coexec
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int KIND>   // 0: f32 16x16x4, 1: f16 16x16x32
__global__ __launch_bounds__(512) void k(int iters, int do_mfma, int do_valu, float* out) {
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    if (wave < 4) {
        if (do_mfma) {
            f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
            float x = threadIdx.x * 1e-3f, y = 1.0f;
            f16x8 hx, hy;
            for (int j = 0; j < 8; ++j) { hx[j] = (_Float16)(x + j); hy[j] = (_Float16)1; }
            for (int i = 0; i < iters; ++i) {
                if (KIND == 0) {
                    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
                    a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
                } else {
                    a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(hx, hy, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(hx, hy, a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(hx, hy, a2, 0, 0, 0);
                    a3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(hx, hy, a3, 0, 0, 0);
                }
            }
            r = a0[0] + a1[1] + a2[2] + a3[3];
        }
    } else if (do_valu) {
        float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f, v4 = 4.f, v5 = 5.f, v6 = 6.f, v7 = 7.f;
        const float m = 1.0000001f, c = 1e-9f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {   // 32 independent-ish FMAs per iteration
                v0 = fmaf(v0, m, c); v1 = fmaf(v1, m, c); v2 = fmaf(v2, m, c); v3 = fmaf(v3, m, c);
                v4 = fmaf(v4, m, c); v5 = fmaf(v5, m, c); v6 = fmaf(v6, m, c); v7 = fmaf(v7, m, c);
            }
        }
        r = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    }
    if (r == 123.456f) out[threadIdx.x] = r;
}

template <int KIND>
float run(int iters, int dm, int dv, float* d) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<KIND><<<256, 512>>>(iters, dm, dv, d);
    hipEventRecord(a);
    k<KIND><<<256, 512>>>(iters, dm, dv, d);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    float* d; hipMalloc(&d, 4096);
    const int iters = 20000;
    for (int kind = 0; kind < 2; ++kind) {
        float tm = kind == 0 ? run<0>(iters, 1, 0, d) : run<1>(iters, 1, 0, d);
        float tv = kind == 0 ? run<0>(iters, 0, 1, d) : run<1>(iters, 0, 1, d);
        float tb = kind == 0 ? run<0>(iters, 1, 1, d) : run<1>(iters, 1, 1, d);
        printf("%s: mfma-only %.3f ms (%.1f cyc/mfma @2.4GHz)  valu-only %.3f ms (%.2f cyc/fma)  both %.3f ms  -> sum %.3f max %.3f\n",
               kind == 0 ? "f32 16x16x4 " : "f16 16x16x32", tm, tm * 2.4e6 / (iters * 4.0), tv, tv * 2.4e6 / (iters * 32.0), tb, tm + tv, tm > tv ? tm : tv);
    }
    return 0;
}
