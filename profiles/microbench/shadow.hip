// shadow.hip - how many filler instructions hide in the shadow of ONE matrix instruction on gfx950, by MFMA shape?
//   shapes : v_mfma_f32_16x16x32_f16 (4 passes, what the edge kernel uses) and v_mfma_f32_32x32x16_f16 (8 passes, same FLOP rate)
//   fillers: v_fma_f32, v_pk_fma_f32, v_exp_f32, v_cvt_pk_f16_f32, ds_read_b128, ds_bpermute_b32
// The loop body is ONE asm volatile block (fixed order: MFMA, NV fillers, MFMA, ...; independent accumulators, fillers on their own
// registers: no hazards), 1 / 2 / 3 waves per SIMD, one workgroup per CU. Reported: shader cycles (s_memtime) per MFMA per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 shadow.hip -o shadow
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// REPn(X) = n filler pairs, alternating two register sets (A: f0 f1 / p0 p1 / d0 d1, B: f4 f5 / p2 p3 / d2 d3): a filler depends on the
// one four instructions before it at the closest
#define REP0(x)
#define REP1(x) x##_A
#define REP2(x) x##_A x##_B
#define REP3(x) x##_A x##_B x##_A
#define REP4(x) x##_A x##_B x##_A x##_B
#define REP6(x) REP4(x) REP2(x)
#define REP8(x) REP4(x) REP4(x)
#define REP12(x) REP8(x) REP4(x)

#define F_FMA_A   "v_fma_f32 %[f0], %[f0], %[m], %[c]\n v_fma_f32 %[f1], %[f1], %[m], %[c]\n"
#define F_FMA_B   "v_fma_f32 %[f4], %[f4], %[m], %[c]\n v_fma_f32 %[f5], %[f5], %[m], %[c]\n"
#define F_PK_A    "v_pk_fma_f32 %[p0], %[p0], %[pm], %[pm]\n v_pk_fma_f32 %[p1], %[p1], %[pm], %[pm]\n"
#define F_PK_B    "v_pk_fma_f32 %[p2], %[p2], %[pm], %[pm]\n v_pk_fma_f32 %[p3], %[p3], %[pm], %[pm]\n"
#define F_EXP_A   "v_exp_f32 %[f0], %[f0]\n v_exp_f32 %[f1], %[f1]\n"
#define F_EXP_B   "v_exp_f32 %[f4], %[f4]\n v_exp_f32 %[f5], %[f5]\n"
#define F_CVT_A   "v_cvt_pk_f16_f32 %[f0], %[f2], %[f3]\n v_cvt_pk_f16_f32 %[f1], %[f2], %[f3]\n"
#define F_CVT_B   "v_cvt_pk_f16_f32 %[f4], %[f2], %[f3]\n v_cvt_pk_f16_f32 %[f5], %[f2], %[f3]\n"
#define F_LDS_A   "ds_read_b128 %[d0], %[addr]\n ds_read_b128 %[d1], %[addr] offset:4096\n"
#define F_LDS_B   "ds_read_b128 %[d2], %[addr] offset:8192\n ds_read_b128 %[d3], %[addr] offset:12288\n"
#define F_BPERM_A "ds_bpermute_b32 %[f0], %[addr], %[f2]\n ds_bpermute_b32 %[f1], %[addr], %[f3]\n"
#define F_BPERM_B "ds_bpermute_b32 %[f4], %[addr], %[f2]\n ds_bpermute_b32 %[f5], %[addr], %[f3]\n"
// every REPn(F_x) emits 2n fillers: the kernels below place HALF = n "pairs" after each MFMA => NV = 2n fillers per MFMA

#define M16(acc) "v_mfma_f32_16x16x32_f16 %[" #acc "], %[a], %[b], %[" #acc "]\n"
#define M32(acc) "v_mfma_f32_32x32x16_f16 %[" #acc "], %[a], %[b], %[" #acc "]\n"

#define KERNEL(name, MF, ACCT, FILL, REP)                                                                              \
    __global__ __launch_bounds__(768) void name(int iters, float* out, unsigned long long* cyc) {                          \
        __shared__ float lds[8192];  /* 32 KB: the ds_read fillers read offsets 0 .. 12288 + 1 KB */                                                                                        \
        for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i;                                                    \
        __syncthreads();                                                                                                   \
        ACCT x0 = {}, x1 = {}, x2 = {}, x3 = {};                                                                           \
        f16x8 a, b;                                                                                                        \
        for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 1e-3f + j); b[j] = (_Float16)1; }                   \
        float f0 = threadIdx.x, f1 = 1.f, f2 = 2.f, f3 = 3.f, f4 = 4.f, f5 = 5.f, m = 1.0000001f, c = 1e-9f;                                   \
        typedef float f32x2 __attribute__((ext_vector_type(2)));                                                           \
        f32x2 p0 = {1.f, 2.f}, p1 = {3.f, 4.f}, p2 = {5.f, 6.f}, p3 = {7.f, 8.f}, pm = {1.0000001f, 1e-9f};                                                  \
        f32x4 d0 = {}, d1 = {}, d2 = {}, d3 = {};                                                                                            \
        int addr = (threadIdx.x & 63) * 16;                                                                                \
        const unsigned long long t0 = __builtin_readcyclecounter();                                                        \
        for (int i = 0; i < iters; ++i) {                                                                                  \
            asm volatile(MF(x0) REP(FILL) MF(x1) REP(FILL) MF(x2) REP(FILL) MF(x3) REP(FILL) "s_waitcnt lgkmcnt(0)\n"      \
                         : [x0] "+v"(x0), [x1] "+v"(x1), [x2] "+v"(x2), [x3] "+v"(x3), [f0] "+v"(f0), [f1] "+v"(f1), [f4] "+v"(f4), [f5] "+v"(f5), [p2] "+v"(p2), [p3] "+v"(p3), [d2] "=&v"(d2), [d3] "=&v"(d3), \
                           [p0] "+v"(p0), [p1] "+v"(p1), [d0] "=&v"(d0), [d1] "=&v"(d1)                                    \
                         : [a] "v"(a), [b] "v"(b), [m] "v"(m), [c] "v"(c), [pm] "v"(pm), [f2] "v"(f2), [f3] "v"(f3), [addr] "v"(addr) \
                         : "memory");                                                                                      \
        }                                                                                                                  \
        asm volatile("s_nop 15\n s_nop 15\n" ::: "memory");                                                                \
        const unsigned long long t1 = __builtin_readcyclecounter();                                                        \
        float r = x0[0] + x1[1] + x2[2] + x3[3] + f0 + f1 + f4 + f5 + p0[0] + p1[1] + p2[0] + p3[1] + d0[0] + d1[1] + d2[2] + d3[3];                                 \
        if (r == 123.456f) out[threadIdx.x] = r;                                                                           \
        if ((threadIdx.x & 63) == 0) atomicAdd(cyc, t1 - t0);                                                              \
    }

#define FAMILY(tag, FILL)                                                     \
    KERNEL(k16_##tag##_0, M16, f32x4, FILL, REP0) KERNEL(k16_##tag##_2, M16, f32x4, FILL, REP1) KERNEL(k16_##tag##_4, M16, f32x4, FILL, REP2) \
    KERNEL(k16_##tag##_6, M16, f32x4, FILL, REP3) KERNEL(k16_##tag##_8, M16, f32x4, FILL, REP4) KERNEL(k16_##tag##_12, M16, f32x4, FILL, REP6) \
    KERNEL(k32_##tag##_0, M32, f32x16, FILL, REP0) KERNEL(k32_##tag##_2, M32, f32x16, FILL, REP1) KERNEL(k32_##tag##_4, M32, f32x16, FILL, REP2) \
    KERNEL(k32_##tag##_6, M32, f32x16, FILL, REP3) KERNEL(k32_##tag##_8, M32, f32x16, FILL, REP4) KERNEL(k32_##tag##_12, M32, f32x16, FILL, REP6) \
    KERNEL(k32_##tag##_16, M32, f32x16, FILL, REP8) KERNEL(k32_##tag##_24, M32, f32x16, FILL, REP12)

FAMILY(fma, F_FMA)
FAMILY(pk, F_PK)
FAMILY(exp, F_EXP)
FAMILY(cvt, F_CVT)
FAMILY(lds, F_LDS)
FAMILY(bperm, F_BPERM)

typedef void (*kern_t)(int, float*, unsigned long long*);
static void run(const char* name, kern_t k, int nv, float* d, unsigned long long* dc) {
    const int iters = 4000;
    printf("%-14s NV=%2d :", name, nv);
    for (int wps = 1; wps <= 3; ++wps) {
        const int threads = 256 * wps;
        hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, 10, d, dc);
        hipMemset(dc, 0, 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, iters, d, dc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long cyc; hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost);
        const double per_wave = (double)cyc / (256.0 * 4 * wps);            // s_memtime ticks per wave (100 MHz ticks or shader cycles: see ratio)
        const double cyc_per_mfma_simd_evt = ms * 1e-3 * 2.4e9 / (iters * 4.0 * wps);
        printf("  %dw/SIMD %6.1f cyc/MFMA/SIMD (evt@2.4GHz; memtime ratio %.3f)", wps, cyc_per_mfma_simd_evt, per_wave / (ms * 1e-3 * 2.4e9));
    }
    printf("\n");
}
#define RUNF(tag) \
    run("16x16x32 " #tag, k16_##tag##_0, 0, d, dc); run("16x16x32 " #tag, k16_##tag##_2, 2, d, dc); run("16x16x32 " #tag, k16_##tag##_4, 4, d, dc); \
    run("16x16x32 " #tag, k16_##tag##_6, 6, d, dc); run("16x16x32 " #tag, k16_##tag##_8, 8, d, dc); run("16x16x32 " #tag, k16_##tag##_12, 12, d, dc); \
    run("32x32x16 " #tag, k32_##tag##_0, 0, d, dc); run("32x32x16 " #tag, k32_##tag##_2, 2, d, dc); run("32x32x16 " #tag, k32_##tag##_4, 4, d, dc); \
    run("32x32x16 " #tag, k32_##tag##_6, 6, d, dc); run("32x32x16 " #tag, k32_##tag##_8, 8, d, dc); run("32x32x16 " #tag, k32_##tag##_12, 12, d, dc); \
    run("32x32x16 " #tag, k32_##tag##_16, 16, d, dc); run("32x32x16 " #tag, k32_##tag##_24, 24, d, dc);
int main() {
    float* d; hipMalloc(&d, 1 << 16);
    unsigned long long* dc; hipMalloc(&dc, 8);
    RUNF(fma) RUNF(pk) RUNF(exp) RUNF(cvt) RUNF(lds) RUNF(bperm)
    return 0;
}
