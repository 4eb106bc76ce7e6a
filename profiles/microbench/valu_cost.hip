// valu_cost.hip - issue cost of single VALU opcodes on gfx950 relative to v_fma_f32: 64 independent instructions of one kind per loop
// iteration, three waves per SIMD, one workgroup per CU. Reported: wall-clock ns per instruction per SIMD and the ratio to v_fma_f32.
// build: hipcc --offload-arch=gfx950 -O3 valu_cost.hip -o valu_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#define R4(x) x x x x
#define R16(x) R4(x) R4(x) R4(x) R4(x)
#define R32(x) R16(x) R16(x)
#define KERNEL(name, BODY)                                                                                         \
    __global__ __launch_bounds__(768) void name(int iters, float* out) {                                           \
        float f0 = threadIdx.x, f1 = 1.f, f2 = 2.f, f3 = 3.f;                                                      \
        int i0 = threadIdx.x, i1 = 3;                                                                              \
        long long l0 = threadIdx.x, l1 = 77, l2 = 5;                                                               \
        typedef float f32x2 __attribute__((ext_vector_type(2)));                                                   \
        f32x2 p0 = {1.f, 2.f}, p1 = {3.f, 4.f};                                                                    \
        for (int i = 0; i < iters; ++i) {                                                                          \
            asm volatile(R32(BODY)                                                                                 \
                         : [f0] "+v"(f0), [f1] "+v"(f1), [i0] "+v"(i0), [l0] "+v"(l0), [l2] "+v"(l2), [p0] "+v"(p0), [p1] "+v"(p1)  \
                         : [f2] "v"(f2), [f3] "v"(f3), [i1] "v"(i1), [l1] "v"(l1) : "vcc", "s10", "s11", "memory");                     \
        }                                                                                                          \
        float r = f0 + f1 + i0 + (float)l0 + (float)l2 + p0[0] + p1[1];                                            \
        if (r == 123.456f) out[threadIdx.x] = r;                                                                   \
    }
KERNEL(k_fma,      "v_fma_f32 %[f0], %[f0], %[f2], %[f3]\n v_fma_f32 %[f1], %[f1], %[f2], %[f3]\n")
KERNEL(k_pkfma,    "v_pk_fma_f32 %[p0], %[p0], %[p0], %[p0]\n v_pk_fma_f32 %[p1], %[p1], %[p1], %[p1]\n")
KERNEL(k_exp,      "v_exp_f32 %[f0], %[f0]\n v_exp_f32 %[f1], %[f1]\n")
KERNEL(k_exp_fma,  "v_exp_f32 %[f0], %[f0]\n v_fma_f32 %[f1], %[f1], %[f2], %[f3]\n")
KERNEL(k_exp_3fma, "v_exp_f32 %[f0], %[f0]\n v_fma_f32 %[f1], %[f1], %[f2], %[f3]\n v_fma_f32 %[f1], %[f1], %[f2], %[f3]\n v_fma_f32 %[f1], %[f1], %[f2], %[f3]\n")
KERNEL(k_med3,     "v_med3_f32 %[f0], %[f0], %[f2], %[f3]\n v_med3_f32 %[f1], %[f1], %[f2], %[f3]\n")
KERNEL(k_cvtpk,    "v_cvt_pk_f16_f32 %[f0], %[f2], %[f3]\n v_cvt_pk_f16_f32 %[f1], %[f2], %[f3]\n")
KERNEL(k_mixlo,    "v_fma_mixlo_f16 %[f0], %[f2], %[f3], %[f0] op_sel_hi:[0,0,0]\n v_fma_mixhi_f16 %[f1], %[f2], %[f3], %[f1] op_sel_hi:[0,0,0]\n")
KERNEL(k_mad64,    "v_mad_i64_i32 %[l0], s[10:11], %[i1], %[i1], %[l1]\n v_mad_i64_i32 %[l2], s[10:11], %[i1], %[i1], %[l1]\n")
KERNEL(k_lshladd64,"v_lshl_add_u64 %[l0], %[l1], 3, %[l0]\n v_lshl_add_u64 %[l2], %[l1], 3, %[l2]\n")
KERNEL(k_addu32,   "v_add_u32 %[i0], %[i0], %[i1]\n v_add_u32 %[f1], %[f1], %[i1]\n")
KERNEL(k_mul24,    "v_mad_u32_u24 %[i0], %[i0], %[i1], %[i1]\n v_mad_u32_u24 %[f1], %[f1], %[i1], %[i1]\n")
KERNEL(k_mullo,    "v_mul_lo_u32 %[i0], %[i0], %[i1]\n v_mul_lo_u32 %[f1], %[f1], %[i1]\n")
KERNEL(k_movdpp,   "v_mov_b32_dpp %[f0], %[f2] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %[f1], %[f3] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
KERNEL(k_adddpp,   "v_add_f32_dpp %[f0], %[f2], %[f0] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %[f1], %[f3], %[f1] row_half_mirror row_mask:0xf bank_mask:0xf\n")
KERNEL(k_rcp,      "v_rcp_f32 %[f0], %[f0]\n v_rcp_f32 %[f1], %[f1]\n")
KERNEL(k_readlane, "v_readlane_b32 s10, %[f0], 3\n v_readlane_b32 s11, %[f1], 5\n")
KERNEL(k_cndmask,  "v_cndmask_b32 %[f0], %[f2], %[f3], vcc\n v_cndmask_b32 %[f1], %[f2], %[f3], vcc\n")
KERNEL(k_cndmask_s,"v_cndmask_b32_e64 %[f0], %[f2], %[f3], s[10:11]\n v_cndmask_b32_e64 %[f1], %[f2], %[f3], s[10:11]\n")
KERNEL(k_cndmask_d,"v_cndmask_b32 %[f0], %[f0], %[f3], vcc\n v_cndmask_b32 %[f1], %[f1], %[f3], vcc\n")
KERNEL(k_and,      "v_and_b32 %[i0], %[i0], %[i1]\n v_and_b32 %[f1], %[f1], %[i1]\n")
KERNEL(k_bfi,      "v_bfi_b32 %[f0], %[i1], %[f2], %[f0]\n v_bfi_b32 %[f1], %[i1], %[f3], %[f1]\n")
KERNEL(k_mulf,     "v_mul_f32 %[f0], %[f0], %[f2]\n v_mul_f32 %[f1], %[f1], %[f3]\n")
KERNEL(k_max,      "v_max_f32 %[f0], %[f0], %[f2]\n v_max_f32 %[f1], %[f1], %[f3]\n")
KERNEL(k_pkmul,    "v_pk_mul_f32 %[p0], %[p0], %[p1]\n v_pk_add_f32 %[p1], %[p1], %[p0]\n")
KERNEL(k_pkaddf16, "v_pk_add_f16 %[f0], %[f0], %[f2]\n v_pk_fma_f16 %[f1], %[f1], %[f2], %[f3]\n")
// round 6: the f16 hi/lo split and its alternatives (f2, f3 = the two fp32 inputs of a pair; results in f0 / f1)
KERNEL(k_mixlo_nodep, "v_fma_mixlo_f16 %[f0], %[f2], %[f3], %[f3] op_sel_hi:[0,0,0]\n v_fma_mixhi_f16 %[f1], %[f2], %[f3], %[f3] op_sel_hi:[0,0,0]\n")
KERNEL(k_mix_f32,  "v_fma_mix_f32 %[f0], %[f2], %[f3], %[f0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %[f1], %[f2], %[f3], %[f1] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n")
KERNEL(k_cvt_f32_f16, "v_cvt_f32_f16 %[f0], %[f2]\n v_cvt_f32_f16_sdwa %[f1], %[f3] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n")
KERNEL(k_split_cur, "v_cvt_pk_f16_f32 %[f0], %[f2], %[f3]\n v_fma_mixlo_f16 %[f1], %[f0], -1.0, %[f2] op_sel_hi:[1,0,0]\n v_fma_mixhi_f16 %[f1], %[f0], -1.0, %[f3] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n")
KERNEL(k_split_mixf32, "v_cvt_pk_f16_f32 %[f0], %[f2], %[f3]\n v_fma_mix_f32 %[i0], %[f0], -1.0, %[f2] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %[f1], %[f0], -1.0, %[f3] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_cvt_pk_f16_f32 %[f1], %[i0], %[f1]\n")
KERNEL(k_split_cvt, "v_cvt_pk_f16_f32 %[f0], %[f2], %[f3]\n v_cvt_f32_f16 %[i0], %[f0]\n v_cvt_f32_f16_sdwa %[f1], %[f0] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_sub_f32 %[i0], %[f2], %[i0]\n v_sub_f32 %[f1], %[f3], %[f1]\n v_cvt_pk_f16_f32 %[f1], %[i0], %[f1]\n")
typedef void (*kern_t)(int, float*);
static double run(kern_t k, float* d) {
    const int iters = 20000;
    hipLaunchKernelGGL(k, dim3(256), dim3(768), 0, 0, 10, d);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(768), 0, 0, iters, d);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6 / ((double)iters * 64 * 3);     // ns per instruction per SIMD (3 waves per SIMD)
}
int main() {
    float* d; hipMalloc(&d, 1 << 16);
    const double base = run(k_fma, d);
#define SHOW(k) { const double t = run(k, d); printf("%-14s %6.3f ns per instruction per SIMD   x%.2f of v_fma_f32\n", #k, t, t / base); }
    SHOW(k_fma) SHOW(k_pkfma) SHOW(k_pkmul) SHOW(k_exp) SHOW(k_exp_fma) SHOW(k_exp_3fma) SHOW(k_rcp) SHOW(k_med3) SHOW(k_cvtpk) SHOW(k_mixlo) SHOW(k_mad64) SHOW(k_lshladd64)
    SHOW(k_addu32) SHOW(k_mul24) SHOW(k_mullo) SHOW(k_movdpp) SHOW(k_adddpp) SHOW(k_readlane) SHOW(k_cndmask) SHOW(k_cndmask_s) SHOW(k_cndmask_d) SHOW(k_and) SHOW(k_bfi) SHOW(k_mulf) SHOW(k_max) SHOW(k_pkaddf16)
    SHOW(k_mixlo_nodep) SHOW(k_mix_f32) SHOW(k_cvt_f32_f16)
    printf("-- sequences per PAIR of values (ns per instruction; x instructions = per pair): split_cur 3 instr, split_mixf32 4, split_cvt 6\n");
    SHOW(k_split_cur) SHOW(k_split_mixf32) SHOW(k_split_cvt)
    return 0;
}
