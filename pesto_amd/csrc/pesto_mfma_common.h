// pesto_mfma_common.h - device helpers shared by the node kernels (pesto_node.hip) and the edge kernel (pesto_edge.hip): vector types, the
// f16 hi/lo split, the range guard of the split path, ELU forms, MFMA block helpers, DPP reductions, the timing-only ablation switches.
//
// MFMA conventions (16x16x4 f32): lane l = (c = l & 15, g = l >> 4).  D[4g + r][c] is register r of lane l.
// Operands chain without shuffles: a D tile of features (rows 16fb + 4g + r) x edges (cols c) is fed back as the
// B operand (or as the A operand, edges as rows) of the next layer with k-step (fb, r) carrying feature
// 16fb + 4g + r from lane group g; weight fragments are stored to match: lane (o, kg) holds W[o][16fb + 4kg + r].
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "pesto_kernels.h"

namespace pesto {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));

// fp32 -> (hi, lo) f16 pair with x = hi + lo to ~2^-21 relative: hi = rn16(x), lo = rn16(x - hi) (the residual is
// exact in fp32). Two 16-feature blocks (4 + 4 values of this lane) form the 8 k-values one lane feeds to
// v_mfma_f32_16x16x32_f16. Three MFMAs (hi*hi, lo*hi, hi*lo) then reproduce the fp32 product to ~2^-21.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// The residual lo = rn16(x - hi) is ONE mixed-precision fma per element (v_fma_mixlo_f16 / v_fma_mixhi_f16: f16(f16(hi) * -1 + x),
// the same single rounding since x - hi is exact in fp32): 12 VALU per 8 values instead of 20 for convert-back / subtract /
// convert. The compiler only selects the mix forms when the multiplier is not a foldable constant (fma(h, -1, x) is
// canonicalised to a subtract first), hence the opaque scalar -1. Compiler-generated, so the MFMA hazard recogniser sees the
// instructions - round 1's inline-asm version of the same idea was invisible to it and corrupted an in-flight SrcC.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split8(f32x4 a, f32x4 b, f16x8& hi, f16x8& lo) {
    const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    float m1 = -1.0f;
    asm("" : "+s"(m1));
#pragma unroll
    for (int j = 0; j < 8; j += 2) {   // pairs, round to nearest (v_cvt_pk_f16_f32 on gfx950): |x - hi| <= 2^-12 |x|, hi + lo carries 2^-22
        const f32x2 x = {v[j], v[j + 1]};
        const f16x2 h = __builtin_convertvector(x, f16x2);
        hi[j] = h[0]; hi[j + 1] = h[1];
#ifdef PESTO_ABL_NOSPLIT   // ablation: no residual (results wrong): -8 VALU per eight values
        lo[j] = h[0]; lo[j + 1] = h[1];
#else
#ifdef PESTO_SPLIT_MIXLO      // rounds 1 - 5: one v_fma_mix{lo,hi}_f16 per element (2.48 issue units each)
        lo[j] = (_Float16)__builtin_fmaf((float)h[0], m1, v[j]);
        lo[j + 1] = (_Float16)__builtin_fmaf((float)h[1], m1, v[j + 1]);
#else
        // the residuals in fp32 (v_fma_mix_f32 reads the f16 halves directly: 1.34 units each; x - hi is exact in fp32), then ONE
        // v_cvt_pk_f16_f32 for the pair: 5.0 instead of 6.1 issue units per pair (profiles/microbench/r06_valu_cost.txt), the same single
        // rounding to f16 - same bits. (opaque: the compiler would fuse fptrunc(fma(fpext)) back into the f16-destination forms)
        float r0 = __builtin_fmaf((float)h[0], m1, v[j]), r1 = __builtin_fmaf((float)h[1], m1, v[j + 1]);
        asm("" : "+v"(r0));
        asm("" : "+v"(r1));
        const f16x2 l = __builtin_convertvector(f32x2{r0, r1}, f16x2);
        lo[j] = l[0]; lo[j + 1] = l[1];
#endif
#endif
    }
}
// Range guard of the f16-split path. A value beyond +-65504 splits into hi = +-inf, lo = -+inf, and every MFMA output fed by
// it becomes NaN (inf - inf, or 0 * inf). ELU = med3(x, exp(x) - 1, 0) turns that NaN into 0, i.e. into a silently wrong
// result, so one register of every accumulator chain is probed BEFORE its ELU: x * 0 + acc stays 0 for finite x and
// becomes NaN for inf / NaN (one v_fmac_f32). Chains that reach Z or the state without an ELU (keys -> softmax weights,
// values -> weighted sums, qpm / ppm outputs) carry their NaN to the next probes downstream. A wave whose probe ended
// as NaN sets bit 4 of the flags word; the host re-runs the forward on the exact fp32 kernels (PESTO_PRECISION_AUTO) or
// reports the range error, and the pool kernel turns every logit into NaN - never a plausible wrong number.
__device__ __forceinline__ void sat_probe(float& acc, float x) { acc = __builtin_fmaf(x, 0.0f, acc); }
__device__ __forceinline__ void sat_flush(float acc, int* __restrict__ flags) {
    if (acc != acc) atomicOr(flags, 4);
}
// The guard is kept PER STRUCTURE (SatCtx, pesto_kernels.h): a lane's probes cover the MFMA columns of one centre at a time, and NaN /
// inf never cross from one structure of a launch to another (neighbour gathers stay inside a structure; MFMA columns - and, with edges
// as rows, rows - are independent). A probe that ended as NaN sets bit 2 of the launch's flags word AND of the word of the centre's
// structure; the pool kernel turns only that structure's logits into NaN and PESTO_PRECISION_AUTO repeats only that structure in fp32,
// so a structure's bits do not depend on its batch mates. row = index into the state arrays (0 = sink: no structure, never flagged).
// The kernels take only the flags word's address, as before the guard was per structure (their register budget is exhausted: three more
// kernel arguments spilled in the nn = 64 instantiations); the rare path reads the SatCtx the forward's first launch left behind the word.
__device__ __forceinline__ void sat_flush_at(float acc, int* __restrict__ flags, int row) {
    if (acc != acc && row > 0) {
        const SatCtx sc = *reinterpret_cast<const SatCtx*>(flags + SATCTX_OFFSET_INTS);
        atomicOr(flags, 4);
        atomicOr(sc.sflags + (sc.seg_of_atom ? sc.seg_of_atom[row - 1] : sc.frame_n ? (row - 1) / sc.frame_n : 0), 4);
    }
}
// conditioning trigger (SatCtx::state_limit): max |new state| of this lane's centre column (8 values of one role's slice) against the limit;
// seven v_max + a compare per role and 16 centres. The limit is read with the other loads of the finish phase (state_limit_of), well
// ahead of the compare. NaN compares false: overflow stays the probes' business.
__device__ __forceinline__ float state_limit_of(const int* __restrict__ flags) {
    return reinterpret_cast<const SatCtx*>(flags + SATCTX_OFFSET_INTS)->state_limit;
}
__device__ __forceinline__ void mag_flush_at(f32x4 a, f32x4 b, float limit, int* __restrict__ flags, int row) {
    const float m = fmaxf(fmaxf(fmaxf(fabsf(a[0]), fabsf(a[1])), fmaxf(fabsf(a[2]), fabsf(a[3]))),
                          fmaxf(fmaxf(fabsf(b[0]), fabsf(b[1])), fmaxf(fabsf(b[2]), fabsf(b[3]))));
    if (m > limit && row > 0) {
        const SatCtx sc = *reinterpret_cast<const SatCtx*>(flags + SATCTX_OFFSET_INTS);
        atomicOr(flags, 4);
        atomicOr(sc.sflags + (sc.seg_of_atom ? sc.seg_of_atom[row - 1] : sc.frame_n ? (row - 1) / sc.frame_n : 0), 4);
    }
}
// AUTO's fp32 repeat (SatCtx::only_flagged): does any of the n <= 64 atom rows i0 .. i0 + n - 1 belong to a structure whose guard word is set?
// Wave-uniform (one ballot); row 0 is the sink, rows >= N1 do not exist.
__device__ __forceinline__ bool only_flagged_of(const int* __restrict__ flags) {
    return flags && reinterpret_cast<const SatCtx*>(flags + SATCTX_OFFSET_INTS)->only_flagged != 0;
}
__device__ __forceinline__ bool rows_flagged(const int* __restrict__ flags, int i0, int n, int N1, int lane) {
    const SatCtx sc = *reinterpret_cast<const SatCtx*>(flags + SATCTX_OFFSET_INTS);
    const int i = i0 + lane;
    bool f = false;
    if (lane < n && i > 0 && i < N1) f = (sc.sflags[sc.seg_of_atom ? sc.seg_of_atom[i - 1] : sc.frame_n ? (i - 1) / sc.frame_n : 0] & 4) != 0;
    return __ballot(f) != 0;
}
__device__ __forceinline__ f16x8 ld8h(const float* p) { return *reinterpret_cast<const f16x8*>(p); }
#define PESTO_WL(fr) ld8h((fr) + 256)      // the lo fragment follows the hi fragment of the same block
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

#ifdef PESTO_ABL_NOELU
__device__ __forceinline__ float elu_f(float x) { return x; }
#else
// ELU(x) = x > 0 ? x : exp(x) - 1 = med3(x, exp(x) - 1, 0): exp(x) - 1 >= x everywhere, so for x > 0 the order is 0 < x <= em1
// and for x <= 0 it is x <= em1 <= 0. One v_med3_f32 instead of compare + select; exp as v_mul + v_exp_f32 (2^t).
__device__ __forceinline__ float elu_f(float x) {
    return __builtin_amdgcn_fmed3f(x, __builtin_amdgcn_exp2f(x * 1.44269504088896340736f) - 1.0f, 0.0f);
}
#endif
__device__ __forceinline__ f32x4 elu4(f32x4 v) {
#ifdef PESTO_ABL_NOELU
    return v;
#else
    const f32x4 t = v * 1.44269504088896340736f;        // vector form: the scale and the -1 become packed-f32 ops
    f32x4 ex = f32x4{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1]), __builtin_amdgcn_exp2f(t[2]), __builtin_amdgcn_exp2f(t[3])};
    ex = ex - 1.0f;
    return f32x4{__builtin_amdgcn_fmed3f(v[0], ex[0], 0.0f), __builtin_amdgcn_fmed3f(v[1], ex[1], 0.0f),
                 __builtin_amdgcn_fmed3f(v[2], ex[2], 0.0f), __builtin_amdgcn_fmed3f(v[3], ex[3], 0.0f)};
#endif
}
// log2-domain ELU of the f16-split edge MLPs (pesto_schema.cpp): t = log2(e) x in, log2(e) ELU(x) out - the exp is a bare v_exp_f32
// and scale + "-1" collapse into one packed fma: 2.5 VALU + 1 transcendental per value instead of 3 + 1
__device__ __forceinline__ f32x4 elu4s(f32x4 t) {
#ifdef PESTO_ABL_NOELU   // ablation: one v_max per value instead of exp + fma + med3 (keeps the magnitudes of the activations)
    return f32x4{fmaxf(t[0], -1.4426950f), fmaxf(t[1], -1.4426950f), fmaxf(t[2], -1.4426950f), fmaxf(t[3], -1.4426950f)};
#endif
    constexpr float C = 1.44269504088896340736f;
    f32x4 ex = f32x4{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1]), __builtin_amdgcn_exp2f(t[2]), __builtin_amdgcn_exp2f(t[3])};
    ex = ex * C - C;
    return f32x4{__builtin_amdgcn_fmed3f(t[0], ex[0], 0.0f), __builtin_amdgcn_fmed3f(t[1], ex[1], 0.0f),
                 __builtin_amdgcn_fmed3f(t[2], ex[2], 0.0f), __builtin_amdgcn_fmed3f(t[3], ex[3], 0.0f)};
}
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
// Centre records are stored FINITE (v_med3 maps NaN / +-inf to +-3e38): in an nn = 8 tile two centres share one centre MFMA, each seeing
// the other's record against a zero B column - 0 x NaN would carry an overflowed structure's NaN into the first / last atom of its
// neighbour in the launch. The overflow itself has been flagged by the probes in front of the store (per structure), the structure is
// computed again in fp32; what the split kernels go on computing for it no longer matters, but it must stay inside it.
__device__ __forceinline__ void st4_finite(float* p, f32x4 v) {
    constexpr float M = 3.0e38f;
    st4(p, f32x4{__builtin_amdgcn_fmed3f(v[0], -M, M), __builtin_amdgcn_fmed3f(v[1], -M, M), __builtin_amdgcn_fmed3f(v[2], -M, M),
                 __builtin_amdgcn_fmed3f(v[3], -M, M)});
}

// ||p|| over xyz of the f16-split path's node inputs (model_operations.py:105): the bare v_sqrt_f32 (1 ulp). sqrtf() is correctly rounded and
// expands to ~17 VALU per value (scaling of denormal inputs, two Newton corrections, class checks); every node wave / finishing role evaluates
// eight norms per lane for each 16-centre tile - in node-wave mode that was 540 wave-instructions per 16 centres (68 per nn = 8 tile).
// One ulp of ||p|| is 6e-8 relative on an input that is split to 2^-22 anyway. k_node16 and both in-kernel prepare phases share this
// function: their records stay bit-identical to each other. (The exact fp32 kernels keep sqrtf.)
__device__ __forceinline__ float norm3_fast(float x, float y, float z) { return __builtin_amdgcn_sqrtf(x * x + y * y + z * z); }

// acc[m] += W[m-block][fb-block] * x for the four k-steps r of block fb; frag table [m][fb][lane][r] in `wf`
template <int NFB>
__device__ __forceinline__ f32x4 mfma_block(const float* __restrict__ wf, int m, int fb, int lane, f32x4 x, f32x4 acc) {
    const f32x4 w = ld4(wf + ((size_t)(m * NFB + fb) * 64 + lane) * 4);
    acc = MFMA(w[0], x[0], acc);
    acc = MFMA(w[1], x[1], acc);
    acc = MFMA(w[2], x[2], acc);
    acc = MFMA(w[3], x[3], acc);
    return acc;
}

// NM independent accumulators advanced together through the four k-steps of input block fb: consecutive MFMAs
// never depend on each other (a 16x16x4 f32 MFMA issues every 32 cycles but its result is ready after 40)
template <int NM, int NFB>
__device__ __forceinline__ void mfma_multi(const float* __restrict__ wf, int m0, int fb, int lane, f32x4 x, f32x4* acc) {
    f32x4 w[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m) w[m] = ld4(wf + ((size_t)((m0 + m) * NFB + fb) * 64 + lane) * 4);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int m = 0; m < NM; ++m) acc[m] = MFMA(w[m][r], x[r], acc[m]);
}

// NM output blocks advanced through k-group kgp (two 16-feature input blocks) on v_mfma_f32_16x16x32_f16 with both
// operands split into f16 hi/lo pairs: acc += wh*xh + wh*xl + wl*xh  (the dropped wl*xl term is ~2^-22 relative).
// Fragment table layout: [m][kgroup][hi|lo][lane][8 halves] (pesto_schema.cpp::put_frags_f16).
// Layer constants -> LDS: all of a thread's loads of a batch are issued before its first store. Written as a plain loop the copy is load -
// wait - store per 16 bytes: one dependent L2 round trip per pass over the workgroup (six to eight in front of every edge launch's first
// work item, twelve in k_node16: 5,200 - 6,100 cycles by the one-wave timeline; one structure per call 0.832 -> 0.811 ms, profiles/
// r05_prologue_ab.txt). Batches of at most eight loads (32 registers, prologue only).
template <int N4, int NT>
__device__ __forceinline__ void copy_to_lds(f32x4* __restrict__ d4, const f32x4* __restrict__ s4, int tid) {
    constexpr int NIT = (N4 + NT - 1) / NT, BATCH = 8;
#pragma unroll
    for (int j0 = 0; j0 < NIT; j0 += BATCH) {
        f32x4 tmp[BATCH];
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
            const int k = tid + (j0 + j) * NT;
            if (j0 + j < NIT) tmp[j] = s4[k < N4 ? k : N4 - 1];      // (unconditional, and pinned below: a masked last load was sunk behind the other stores)
        }
#pragma unroll
        for (int j = 0; j < BATCH; ++j)
            if (j0 + j < NIT) asm volatile("" : "+v"(tmp[j]));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
            const int k = tid + (j0 + j) * NT;
            if (j0 + j < NIT && k < N4) d4[k] = tmp[j];
        }
    }
}
template <int NM>
__device__ __forceinline__ void mfma16_multi(const float* __restrict__ wf, int m0, int nkg, int kgp, int lane, f16x8 xh, f16x8 xl,
                                             f32x4* acc) {
    f16x8 wh[NM], wl[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        const float* fr = wf + (size_t)(((m0 + m) * nkg + kgp) * 2) * 256 + lane * 4;
        wh[m] = ld8h(fr); wl[m] = PESTO_WL(fr);
    }
#pragma unroll
    for (int m = 0; m < NM; ++m) acc[m] = MFMA16(wh[m], xh, acc[m]);
#pragma unroll
    for (int m = 0; m < NM; ++m) acc[m] = MFMA16(wh[m], xl, acc[m]);
#pragma unroll
    for (int m = 0; m < NM; ++m) acc[m] = MFMA16(wl[m], xh, acc[m]);
}

// ---- buffer addressing: a 128-bit resource (base pointer in SGPRs, no bounds: the range is the whole 4 GB window behind the base) + a
// 32-bit per-lane byte offset + a wave-uniform (SGPR) byte offset + a 12-bit immediate the compiler folds constant parts into. A global
// load needs the full 64-bit address per lane (v_mad_i64_i32 / v_lshl_add_u64 / add_co pairs at 1.4 issue units each); here the per-lane
// part is one 32-bit multiply-add and everything uniform costs scalar instructions only. Every array addressed this way is far below 4 GB
// (the largest, the centre records, 2,112 B per atom: two million atoms).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0xfffffffc, 0x00020000);
}
// AUX = cache-policy bits of the instruction (gfx950: 1 = sc0, 2 = nt, 16 = sc1), 0 = default. Measured for the data a launch reads ONCE
// (centre records, edge rows; profiles/r06_streaming_loads_ab.txt): nt loads are 6.5 % SLOWER on the step, sc1 loads change nothing -
// every load of the layer kernels uses the default policy. (nt pays on the 288 MB dense mask, which is larger than the Infinity Cache.)
template <int AUX = 0>
__device__ __forceinline__ f32x4 bufld4(__amdgpu_buffer_rsrc_t r, int v_off, int s_off = 0) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, v_off, s_off, AUX));
}
__device__ __forceinline__ f16x8 bufld8h(__amdgpu_buffer_rsrc_t r, int v_off, int s_off = 0) {
    return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(r, v_off, s_off, 0));
}
template <int AUX = 0>
__device__ __forceinline__ f32x2 bufld2(__amdgpu_buffer_rsrc_t r, int v_off, int s_off = 0) {
    return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, v_off, s_off, AUX));
}
template <int AUX = 0>
__device__ __forceinline__ float bufld1(__amdgpu_buffer_rsrc_t r, int v_off, int s_off = 0) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, v_off, s_off, AUX));
}
__device__ __forceinline__ void bufst4(__amdgpu_buffer_rsrc_t r, int v_off, int s_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, v_off, s_off, 0);
}
// scalar base pointer + 32-bit per-lane byte offset (+ uniform byte offset): the compiler selects the SADDR form of global_load / global_store
// (base in an SGPR pair, zero-extended VGPR offset, 13-bit immediate) - no 64-bit per-lane address arithmetic and no resource descriptor
__device__ __forceinline__ f32x4 gld4(const void* base, unsigned v_off, unsigned s_off = 0) {
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + s_off + (size_t)v_off);
}
__device__ __forceinline__ void gst4(void* base, unsigned v_off, unsigned s_off, f32x4 v) {
    *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(base) + s_off + (size_t)v_off) = v;
}
// write-through form (global_store ... sc1): the line goes to memory now and is dropped from this XCD's L2 instead of staying dirty until
// the end-of-kernel write-back (MI355X_MICROARCH.md "boundary": + dirty bytes / 6 TB/s behind a kernel that leaves them; "stores of each
// flavour": a 16-byte sc1 store costs what a plain one does). For data nobody reads before the next launch: the new state and the next
// layer's records (76 MB per launch at 24 k atoms) - same box 1,862 -> 1,913 structures/s, nn = 8 -5.7 % per launch, the others -2 %
// (profiles/r06_writethrough_ab.txt). Inline asm: no builtin emits
// the cache-policy bits on a plain global store; the trailing s_nop covers the store-data hazard the compiler cannot see.
__device__ __forceinline__ void gst4_wt(void* base, unsigned v_off, unsigned s_off, f32x4 v) {
    const char* b = reinterpret_cast<const char*>(base) + s_off;
    // (s_nop 4 in front: the SGPR pair may have just been reloaded from a spill lane by v_readlane - VALU writes SGPR -> VMEM reads it needs
    //  five wait states, and the hazard recogniser does not look inside inline asm: without it the store went to a stale base and faulted)
    asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" : : "v"(v_off), "v"(v), "s"(b));
}
__device__ __forceinline__ void st4_wt(float* p, f32x4 v) {      // the same with a per-lane 64-bit address (k_node16)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v));
}
__device__ __forceinline__ void st4_wt_finite(float* p, f32x4 v) {
    constexpr float M = 3.0e38f;
    st4_wt(p, f32x4{__builtin_amdgcn_fmed3f(v[0], -M, M), __builtin_amdgcn_fmed3f(v[1], -M, M), __builtin_amdgcn_fmed3f(v[2], -M, M),
                    __builtin_amdgcn_fmed3f(v[3], -M, M)});
}
__device__ __forceinline__ void gst4_finite(void* base, unsigned v_off, unsigned s_off, f32x4 v) {      // see st4_finite
    constexpr float M = 3.0e38f;
    gst4_wt(base, v_off, s_off, f32x4{__builtin_amdgcn_fmed3f(v[0], -M, M), __builtin_amdgcn_fmed3f(v[1], -M, M), __builtin_amdgcn_fmed3f(v[2], -M, M),
                                            __builtin_amdgcn_fmed3f(v[3], -M, M)});
}
__device__ __forceinline__ void bufst4_finite(__amdgpu_buffer_rsrc_t r, int v_off, int s_off, f32x4 v) {      // see st4_finite
    constexpr float M = 3.0e38f;
    bufst4(r, v_off, s_off, f32x4{__builtin_amdgcn_fmed3f(v[0], -M, M), __builtin_amdgcn_fmed3f(v[1], -M, M), __builtin_amdgcn_fmed3f(v[2], -M, M),
                                  __builtin_amdgcn_fmed3f(v[3], -M, M)});
}

// ---- cross-lane reductions on the VALU (DPP) instead of ds_bpermute round trips through the LDS crossbar
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float lane_bcast(float x, int src_lane) {   // src_lane must be wave-uniform
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), src_lane));
}
// reduce over the 16 lanes of a DPP row (W16) or over each 8-lane half (!W16); every lane ends with the result.
// The permuted operand is a bound_ctrl mov_dpp without an `old` value: the compiler folds it into the add (v_add_f32_dpp, one
// instruction per step instead of v_mov 0 + v_mov_dpp + v_add); the max keeps fmaxf (canonicalisation of the permuted operand + v_max).
template <int CTRL>
__device__ __forceinline__ float dpp_perm(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
template <bool W16, bool IS_MAX>
__device__ __forceinline__ float row_reduce(float x) {
#define PESTO_RR(ctrl) { const float y = dpp_perm<ctrl>(x); x = IS_MAX ? fmaxf(x, y) : x + y; }
    PESTO_RR(0xB1)          // quad_perm [1,0,3,2]
    PESTO_RR(0x4E)          // quad_perm [2,3,0,1]
    PESTO_RR(0x141)         // row_half_mirror: i <-> 7-i
    if (W16) PESTO_RR(0x140)   // row_mirror: i <-> 15-i
#undef PESTO_RR
    return x;
}

// workgroup rendezvous that orders LDS traffic only: __syncthreads() also waits for every outstanding GLOBAL load of the wave
// (vmcnt(0)), which would expose the latency of the loads the finish phase deliberately issues ahead of the rendezvous
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// timing-only ablations of the memory side (results wrong; profiles/ab.sh): PESTO_ABL_NOGATHER = every neighbour gather (A_j, p_j) reads one
// of 8 hot rows; PESTO_ABL_NOCENLD = every centre-record / own-state read hits one of 16 hot records; PESTO_ABL_NOPREPST = the prepare
// phase's record stores alias into 16 records (the stores are issued, the fabric sees 40 KB)
#ifdef PESTO_ABL_NOGATHER
#define ABL_NB(x) ((x) & 7)
#else
#define ABL_NB(x) (x)
#endif
#ifdef PESTO_ABL_NOCENLD
#define ABL_CEN(x) ((x) & 15)
#else
#define ABL_CEN(x) (x)
#endif
#ifdef PESTO_ABL_NOPREPST
#define ABL_ST(x) ((x) & 15)
#else
#define ABL_ST(x) (x)
#endif
}  // namespace pesto
