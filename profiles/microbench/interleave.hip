// interleave.hip - how many independent VALU FMAs hide in the shadow of one v_mfma_f32_16x16x4_f32 issued by the SAME wave?
// One or two waves per SIMD; per loop iteration 4 MFMAs (independent accumulators), each followed by NV v_fma.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NV>
__global__ void k(int iters, float* out) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    float x = threadIdx.x * 1e-3f, y = 1.0f;
    float v[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    const float m = 1.0000001f, c = 1e-9f;
    for (int i = 0; i < iters; ++i) {
#define STEP(acc) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc, 0, 0, 0); \
        _Pragma("unroll") for (int j = 0; j < NV; ++j) v[j & 7] = fmaf(v[j & 7], m, c);
        STEP(a0) STEP(a1) STEP(a2) STEP(a3)
    }
    float r = a0[0] + a1[1] + a2[2] + a3[3];
    for (int j = 0; j < 8; ++j) r += v[j];
    if (r == 123.456f) out[threadIdx.x] = r;
}
template <int NV> void run(int threads, float* d) {
    const int iters = 20000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<NV><<<256, threads>>>(iters, d);
    hipEventRecord(a); k<NV><<<256, threads>>>(iters, d); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("waves/SIMD %d  NV=%2d  %.1f cycles per (MFMA + %d fma) per wave-slot\n", threads / 256, NV, ms * 2.4e6 / (iters * 4.0) / (threads / 256), NV);
}
int main() {
    float* d; hipMalloc(&d, 4096);
    for (int threads : {256, 512}) {
        run<0>(threads, d); run<2>(threads, d); run<4>(threads, d); run<6>(threads, d); run<8>(threads, d); run<12>(threads, d); run<16>(threads, d);
    }
    return 0;
}
