    // measured alternative (round 5, -DPESTO_ELU_MAX): c ELU(x) = max(t, c min(2^t, 1) - c) with the clamp as v_exp_f32's output modifier and one
    // VOP2 v_max_f32 instead of the VOP3 v_med3_f32 - cheaper by the price list of r03_valu_cost.txt, SLOWER in the kernel (same box:
    // nn = 64 260.4 -> 263.4 us, nn = 32 147.0 -> 149.2; the clamp forces the 64-bit encoding of v_exp_f32 and moves the schedule)
// pesto_layer_mfma.hip - the state-update layer on the gfx950 matrix cores.
//
// Reference math: src/model_operations.py:87-154 (StateUpdate.forward) + :225-242 (StateUpdateLayer.forward).
// One layer = two kernels:
//
//   k_node16 per ATOM, batched as MFMA GEMMs over 16-atom column tiles (four waves per tile):
//            (finish) q += qpm(Zq), p += ppm(Zp) of the previous layer, sink row reset            (:147-152, :239-240)
//                     - on the shipped path this half runs INSIDE the edge kernel (FIN, below)
//            (prep)   the first Linear of the three edge MLPs is linear in its 193 inputs
//                     [d | X_n(i) | q_j | |p_j| | p_i.r | p_j.r]  (:109-116), so its per-atom pieces are computed
//                     ONCE per atom instead of once per edge (exact algebra, different summation order):
//                       centre record   U_i = b1 + W[:,1:65] X_n(i),  G_i[c] = W[:,129:161] p_i[c],  Q_i = nqm(X_n(i))
//                       neighbour record A_j = W[:,65:129] X_n(j)     (shipped "hybrid" path; the exact fp32 kernels also store
//                                        C_j[c] = W[:,161:193] p_j[c], a 2 KB record)
//   k_edge   per EDGE: h1 = ELU(U_i + sum_c r_c G_i[c] + w_d d + A_j + W[:,161:193] (p_j . r))  - centre terms by one K = 4
//            fp32 MFMA per block, the p_j . r block per edge on f16-split MFMA from the gathered p_j - then layers 2/3 of
//            eqkm/epkm/evm as MFMA chains held in registers, both softmaxes with DPP reductions, attention-weighted sums
//            Zq/Zp per atom (:119-144). FIN (shipped): the sums stay in LDS; behind a workgroup rendezvous four waves per 16
//            centres apply qpm / ppm + residual on the matrix cores and write the new state into a ping-pong pair.
//   Shipped arithmetic: every large GEMM as f16 hi/lo split on v_mfma_f32_16x16x32_f16 (x.w = xh.wh + xl.wh + xh.wl, fp32
//   accumulate); k_node / k_edge<..., F16 = false> (PESTO_PRECISION_FP32) keep everything on exact fp32 v_mfma_f32_16x16x4_f32.
//
// MFMA conventions (16x16x4 f32): lane l = (c = l & 15, g = l >> 4).  D[4g + r][c] is register r of lane l.
// Operands chain without shuffles: a D tile of features (rows 16fb + 4g + r) x edges (cols c) is fed back as the
// B operand (or as the A operand, edges as rows) of the next layer with k-step (fb, r) carrying feature
// 16fb + 4g + r from lane group g; weight fragments are stored to match: lane (o, kg) holds W[o][16fb + 4kg + r].
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
        lo[j] = (_Float16)__builtin_fmaf((float)h[0], m1, v[j]);
        lo[j + 1] = (_Float16)__builtin_fmaf((float)h[1], m1, v[j + 1]);
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
#ifdef PESTO_ABL_NOWL   // ablation: the low weight fragments are not read from LDS (results wrong): -1/3 of the LDS weight traffic
#define PESTO_WL(fr) ld8h(fr) 
#else
#define PESTO_WL(fr) ld8h((fr) + 256)
#endif
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
#ifdef PESTO_ABL_NOMFMA   // ablation: the MFMA becomes one VALU op keeping the data dependence
#define MFMA(a, b, c) ((c) + (a) * (b))
#else
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#endif

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
#ifndef PESTO_ELU_MAX      // shipped: exp, packed fma, v_med3_f32
    f32x4 ex = f32x4{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1]), __builtin_amdgcn_exp2f(t[2]), __builtin_amdgcn_exp2f(t[3])};
    ex = ex * C - C;
    return f32x4{__builtin_amdgcn_fmed3f(t[0], ex[0], 0.0f), __builtin_amdgcn_fmed3f(t[1], ex[1], 0.0f),
                 __builtin_amdgcn_fmed3f(t[2], ex[2], 0.0f), __builtin_amdgcn_fmed3f(t[3], ex[3], 0.0f)};
#else
    // measured alternative (round 5, -DPESTO_ELU_MAX): c ELU(x) = max(t, c min(2^t, 1) - c), the clamp as v_exp_f32's output modifier and one
    // VOP2 v_max_f32 instead of the VOP3 v_med3_f32 - cheaper by the price list of profiles/microbench/r03_valu_cost.txt, SLOWER in the
    // kernel (same box: nn = 64 260.4 -> 263.4 us, nn = 32 147.0 -> 149.2; the clamp forces the 64-bit encoding of v_exp_f32 and the
    // schedule moves). Same value except for 0 < t < 2^-24, where 2^t rounds to 1: med3(t, 0, 0) = 0, this form returns t.
    f32x4 ex = f32x4{__builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(t[0]), 0.0f, 1.0f), __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(t[1]), 0.0f, 1.0f),
                     __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(t[2]), 0.0f, 1.0f), __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(t[3]), 0.0f, 1.0f)};
    ex = ex * C - C;
    return f32x4{__builtin_fmaxf(t[0], ex[0]), __builtin_fmaxf(t[1], ex[1]), __builtin_fmaxf(t[2], ex[2]), __builtin_fmaxf(t[3], ex[3])};
#endif
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

// =============================================================================================== node kernel
// finish >= 0: apply layer `finish`'s output MLPs to Z and update the state in place (sink reset).
// prep   >= 0: write layer `prep`'s centre / neighbour records from the (updated) state.
// One wave = 16 atoms; lane (e = atom in tile, g).  Weight fragments stream from L2 (shared by all waves).
__global__ __launch_bounds__(256) void k_node(const float* __restrict__ W, LayerW wf_, LayerW wp_, int do_finish, int do_prep,
                                              int N1, float* __restrict__ q_state, float* __restrict__ p_state,
                                              const float* __restrict__ Z, float* __restrict__ rec_nb, float* __restrict__ rec_cen,
                                              const int* __restrict__ flags) {
    const int lane = threadIdx.x & 63, e = lane & 15, g = lane >> 4;
    // same XCD-aware atom partition as the edge kernel: XCD b % 8 owns a contiguous eighth of the 16-atom tiles,
    // so the records it writes are the ones its own L2 will be asked for by the edge kernel's centre reads
    const int n_tiles = (N1 + 15) >> 4, chunk = (n_tiles + 7) >> 3;
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int tile = xcd * chunk + jb * 4 + (threadIdx.x >> 6);
    if (tile >= min(n_tiles, (xcd + 1) * chunk)) return;
    // AUTO's fp32 repeat: tiles without an atom of a flagged structure are skipped (tile 0 never: it holds the sink row, whose records
    // every padded neighbour slot gathers)
    if (tile > 0 && only_flagged_of(flags) && !rows_flagged(flags, tile * 16, 16, N1, lane)) return;
    const int i_raw = tile * 16 + e;
    const bool valid = i_raw < N1;
    const int i = valid ? i_raw : N1 - 1;

    // state in B/D layout: q[m] = q[i][16m + 4g .. +3], p[c][m] likewise
    f32x4 q[2], p[3][2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        q[m] = ld4(q_state + (size_t)i * S + 16 * m + 4 * g);
#pragma unroll
        for (int c = 0; c < 3; ++c) p[c][m] = ld4(p_state + (size_t)i * 96 + c * 32 + 16 * m + 4 * g);
    }

    if (do_finish) {
        const float* zr = Z + (size_t)i * REC_Z;
        // qpm: 64 -> 32 -> 32 -> 32 with ELU between            (model_operations.py:147)
        f32x4 h[2], t[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) h[m] = ld4(W + wf_.n_bq0 + 16 * m + 4 * g);
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) mfma_multi<2, 4>(W + wf_.n_q0, 0, fb, lane, ld4(zr + 16 * fb + 4 * g), h);
#pragma unroll
        for (int m = 0; m < 2; ++m) { h[m] = elu4(h[m]); t[m] = ld4(W + wf_.n_bq1 + 16 * m + 4 * g); }
#pragma unroll
        for (int fb = 0; fb < 2; ++fb) mfma_multi<2, 2>(W + wf_.n_q1, 0, fb, lane, h[fb], t);
#pragma unroll
        for (int m = 0; m < 2; ++m) { t[m] = elu4(t[m]); h[m] = ld4(W + wf_.n_bq2 + 16 * m + 4 * g); }
#pragma unroll
        for (int fb = 0; fb < 2; ++fb) mfma_multi<2, 2>(W + wf_.n_q2, 0, fb, lane, t[fb], h);
#pragma unroll
        for (int m = 0; m < 2; ++m) q[m] += h[m];                                                  // :151
        // ppm: 64 -> 32, no bias, per xyz component                                              // :148, :152
        {
            f32x4 a[3][2];
#pragma unroll
            for (int c = 0; c < 3; ++c) { a[c][0] = f32x4{0, 0, 0, 0}; a[c][1] = f32x4{0, 0, 0, 0}; }
#pragma unroll
            for (int fb = 0; fb < 4; ++fb) {
                const f32x4 w0 = ld4(W + wf_.n_pp + ((size_t)(0 * 4 + fb) * 64 + lane) * 4);
                const f32x4 w1 = ld4(W + wf_.n_pp + ((size_t)(1 * 4 + fb) * 64 + lane) * 4);
                f32x4 x[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) x[c] = ld4(zr + 64 + c * 64 + 16 * fb + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) { a[c][0] = MFMA(w0[r], x[c][r], a[c][0]); a[c][1] = MFMA(w1[r], x[c][r], a[c][1]); }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) { p[c][0] += a[c][0]; p[c][1] += a[c][1]; }
        }
        if (i == 0) {                                                                              // :239-240 sink
#pragma unroll
            for (int m = 0; m < 2; ++m) { q[m] = f32x4{0, 0, 0, 0}; p[0][m] = q[m]; p[1][m] = q[m]; p[2][m] = q[m]; }
        }
        if (valid) {
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                st4(q_state + (size_t)i * S + 16 * m + 4 * g, q[m]);
#pragma unroll
                for (int c = 0; c < 3; ++c) st4(p_state + (size_t)i * 96 + c * 32 + 16 * m + 4 * g, p[c][m]);
            }
        }
    }
    if (!do_prep) return;

    // X_n = [q | |p|]  as four 16-feature blocks                                                  // :103-106
    f32x4 xn[4];
    xn[0] = q[0]; xn[1] = q[1];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            xn[2 + m][r] = sqrtf(p[0][m][r] * p[0][m][r] + p[1][m][r] * p[1][m][r] + p[2][m][r] * p[2][m][r]);

    float* cen = rec_cen + (size_t)i * REC_CEN;
    float* nb = rec_nb + (size_t)i * REC_NB;
    // [U | A] = [W[:,1:65]; W[:,65:129]] X_n   (16 output blocks; U carries b1)
#pragma unroll 1
    for (int ob = 0; ob < 16; ob += 4) {
        f32x4 a[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = ob < 8 ? ld4(W + wp_.n_b1 + 16 * (ob + j) + 4 * g) : f32x4{0, 0, 0, 0};
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) mfma_multi<4, 4>(W + wp_.n_ua, ob, fb, lane, xn[fb], a);
        if (valid) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (ob < 8) st4(cen + (ob + j) * 64 + 3 * 16 + 4 * g, a[j]);        // centre record slot kg = 3 (U)
                else st4(nb + ((ob + j - 8) * 4 + g) * 16, a[j]);                  // neighbour record array 0 (A)
            }
        }
    }
    // [G_c | C_c] = [W[:,129:161]; W[:,161:193]] p[c]
#pragma unroll 1
    for (int ob = 0; ob < 16; ob += 2) {
        f32x4 a[2][3];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int c = 0; c < 3; ++c) a[j][c] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int fb = 0; fb < 2; ++fb) {
            const f32x4 w0 = ld4(W + wp_.n_gc + ((size_t)((ob + 0) * 2 + fb) * 64 + lane) * 4);
            const f32x4 w1 = ld4(W + wp_.n_gc + ((size_t)((ob + 1) * 2 + fb) * 64 + lane) * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) { a[0][c] = MFMA(w0[r], p[c][fb][r], a[0][c]); a[1][c] = MFMA(w1[r], p[c][fb][r], a[1][c]); }
        }
        if (valid) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    if (ob < 8) st4(cen + (ob + j) * 64 + c * 16 + 4 * g, a[j][c]);
                    else st4(nb + ((ob + j - 8) * 4 + g) * 16 + (1 + c) * 4, a[j][c]);
                }
        }
    }
    // node queries Q = nqm(X_n): 64 -> 32 -> 32 -> 12 (rows 12..15 of the last block are zero padding)   // :119
    {
        f32x4 h[2], t[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) h[m] = ld4(W + wp_.n_bn0 + 16 * m + 4 * g);
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) mfma_multi<2, 4>(W + wp_.n_n0, 0, fb, lane, xn[fb], h);
#pragma unroll
        for (int m = 0; m < 2; ++m) { h[m] = elu4(h[m]); t[m] = ld4(W + wp_.n_bn1 + 16 * m + 4 * g); }
#pragma unroll
        for (int fb = 0; fb < 2; ++fb) mfma_multi<2, 2>(W + wp_.n_n1, 0, fb, lane, h[fb], t);
        f32x4 qq = ld4(W + wp_.n_bn2 + 4 * g);
#pragma unroll
        for (int fb = 0; fb < 2; ++fb) qq = mfma_block<2>(W + wp_.n_n2, 0, fb, lane, elu4(t[fb]), qq);
        if (valid) st4(cen + 512 + 4 * g, qq);
    }
    // (p_j for the neighbours' vector-value gather is read from the state array itself: it is only rewritten by the
    // NEXT node kernel, after this layer's edge kernel has finished)
}

// NM output blocks advanced through k-group kgp (two 16-feature input blocks) on v_mfma_f32_16x16x32_f16 with both
// operands split into f16 hi/lo pairs: acc += wh*xh + wh*xl + wl*xh  (the dropped wl*xl term is ~2^-22 relative).
// Fragment table layout: [m][kgroup][hi|lo][lane][8 halves] (pesto_schema.cpp::put_frags_f16).
// Layer constants -> LDS: all of a thread's loads of a batch are issued before its first store. Written as a plain loop the copy is load -
// wait - store per 16 bytes: one dependent L2 round trip per pass over the workgroup (six to eight in front of every edge launch's first
// work item, twelve in k_node16: 5,200 - 6,100 cycles by the one-wave timeline; one structure per call 0.832 -> 0.811 ms, profiles/
// r05_prologue_ab.txt). Batches of at most eight loads (32 registers, prologue only). -DPESTO_SERIAL_PROLOGUE builds the plain loop.
template <int N4, int NT>
__device__ __forceinline__ void copy_to_lds(f32x4* __restrict__ d4, const f32x4* __restrict__ s4, int tid) {
#ifndef PESTO_SERIAL_PROLOGUE
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
#else
    for (int k = tid; k < N4; k += NT) d4[k] = s4[k];
#endif
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

// node kernel on the f16-split MFMA path (same contract as k_node; 249 f16 MFMAs per 16 atoms on the hybrid path, 321 with
// full neighbour records, instead of 856 fp32 MFMAs).
// The kernel is LATENCY-bound, not throughput-bound: 24k atoms are only 1,500 wave tiles for 1,024 SIMDs, so its duration is
// the length of one wave's dependent chain. All weight fragments of both halves (finish 24 KB, [U|A] 64 KB, [G|C] 16/32 KB,
// nqm 14 KB) are therefore staged into LDS by ONE fill per workgroup (eight waves, one workgroup per CU) that overlaps the
// state / Z loads - three sequential fill-barrier-compute phases per workgroup cost 27 us per launch, 15 % of a forward.
// Hybrid edge kernel: the neighbour record is A_j[128] in natural feature order (REC_A floats per atom); the
// C_j[c] = W[:,161:193] p_j[c] pieces are no longer materialised - the edge kernel applies that block per edge on the matrix
// cores from the gathered p_j (4x less gather traffic than the 2 KB record, which was the edge kernel's bottleneck).
constexpr int NODE_WAVES = 8;
constexpr int NL_FIN = 0, NL_UA = 6144, NL_GC = NL_UA + 16384;
constexpr int NL_NQ = NL_GC + 4096;
constexpr int NODE_LDS_FLOATS = NL_NQ + 3584;

// Four waves share one 16-atom tile (role = wave & 3) so that the dependent chain a wave walks is ~70 MFMAs, not 249:
//   finish: role 0 -> qpm (q update), roles 1..3 -> ppm for xyz component role - 1; the updated tile state is exchanged through
//           LDS (8 KB per tile) behind one workgroup barrier;
//   prep  : role r -> [U|A] output blocks 4r..4r+3 and G blocks 2r, 2r+1; role 3 also nqm.
// Workgroups are persistent (eight waves = two tiles per iteration, weights resident in LDS).
__global__ __launch_bounds__(NODE_WAVES * 64, 1) void k_node16(const float* __restrict__ W, LayerW wf_, LayerW wp_, int do_finish, int do_prep,
                                                int N1, float* __restrict__ q_state, float* __restrict__ p_state,
                                                const float* __restrict__ Z, float* __restrict__ rec_nb, float* __restrict__ rec_cen,
                                                int* __restrict__ flags, Unpack2Args u2, int n_node_blocks) {
    if ((int)blockIdx.x >= n_node_blocks) {
        // extra workgroups of a small launch: pass 2 of the geometry (k_unpack2's statement, the same expressions: same bits) - it only
        // depends on the launch in front (max(D)), like the records this launch writes only depend on the embedding
        const int64_t n_slots = (int64_t)(u2.n + 1) * KMAX;
        for (int64_t s_ = (int64_t)(blockIdx.x - n_node_blocks) * (NODE_WAVES * 64) + threadIdx.x; s_ < n_slots;
             s_ += (int64_t)(gridDim.x - n_node_blocks) * (NODE_WAVES * 64)) {
            if (s_ < KMAX) { u2.ids_s[s_] = 0; u2.geo[s_] = make_float4(0.f, 0.f, 0.f, 0.f); continue; }
            const float dmax = __uint_as_float(u2.dmax_bits[u2.seg_of_atom ? u2.seg_of_atom[(s_ >> 6) - 1] : ((s_ >> 6) - 1) / u2.Nf]);
            float4 gg = u2.geo[s_];
            const float d = gg.w + dmax * (gg.w < 1e-2f ? 1.0f : 0.0f);
            u2.geo[s_] = make_float4(gg.x / d, gg.y / d, gg.z / d, d);
        }
        return;
    }
    const int lane = threadIdx.x & 63, e = lane & 15, g = lane >> 4;
    float sat = 0.0f;
    // wave-uniform by construction; readfirstlane makes it uniform for the compiler too (scalar branches around the MFMA blocks)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), role = wave & 3, slot = wave >> 2;
    __shared__ __attribute__((aligned(16))) float wl_[NODE_LDS_FLOATS];
    __shared__ __attribute__((aligned(16))) float xch[2][8][256];     // [tile slot][q0 q1 p00 p01 p10 p11 p20 p21][lane][4]
    {   // one fill: [q0 | q1 | q2 | pp] (contiguous in the image), [U|A], G (first half of the [G|C] table), [n0 | n1 | n2]
#define PESTO_NODE_COPY(dst, src, n_floats) copy_to_lds<(n_floats) / 4, NODE_WAVES * 64>(reinterpret_cast<f32x4*>(wl_ + (dst)), reinterpret_cast<const f32x4*>(src), (int)threadIdx.x)
        if (do_finish) PESTO_NODE_COPY(NL_FIN, W + wf_.h_q0, 6144);
        if (do_prep) {
            PESTO_NODE_COPY(NL_UA, W + wp_.h_ua, 16384);
            PESTO_NODE_COPY(NL_GC, W + wp_.h_gc, 4096);
            PESTO_NODE_COPY(NL_NQ, W + wp_.h_n0, 3584);
        }
#undef PESTO_NODE_COPY
    }
    // XCD-aware partition of tile PAIRS (same atom ranges per XCD as the edge kernel's work items)
    const int n_tiles = (N1 + 15) >> 4, n_pairs = (n_tiles + 1) >> 1, chunk = (n_pairs + 7) >> 3;
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3, nbx = n_node_blocks >> 3;
    const int p_end = min(n_pairs, (xcd + 1) * chunk);
    bool first = true;
    for (int pair = xcd * chunk + jb; pair < p_end; pair += nbx) {
        const int tile = 2 * pair + slot;
        const bool live = tile < n_tiles;
        const int i_raw = (live ? tile : 0) * 16 + e;
        const bool valid = live && i_raw < N1;
        const int i = (i_raw < N1) ? i_raw : N1 - 1;
        float* xs = &xch[slot][0][0];

        // this role's slice of the state: role 0 -> q, role c + 1 -> p[c]
        const float* src = role == 0 ? q_state + (size_t)i * S : p_state + (size_t)i * 96 + (role - 1) * 32;
        f32x4 st[2] = {ld4(src + 4 * g), ld4(src + 16 + 4 * g)};
        f16x8 xh, xl;
        if (first) { __syncthreads(); first = false; }      // weights are in LDS
        if (do_finish) {
            const float* Lq0 = wl_ + NL_FIN, *Lq1 = Lq0 + 2048, *Lq2 = Lq0 + 3072, *Lpp = Lq0 + 4096;
            const float* zr = Z + (size_t)i * REC_Z;
            if (role == 0) {   // qpm: 64 -> 32 -> 32 -> 32 with ELU between            (model_operations.py:147, :151)
                f32x4 h[2], t[2];
#pragma unroll
                for (int m = 0; m < 2; ++m) h[m] = ld4(W + wf_.n_bq0 + 16 * m + 4 * g);
#pragma unroll
                for (int kgp = 0; kgp < 2; ++kgp) {
                    split8(ld4(zr + 32 * kgp + 4 * g), ld4(zr + 32 * kgp + 16 + 4 * g), xh, xl);
                    mfma16_multi<2>(Lq0, 0, 2, kgp, lane, xh, xl, h);
                }
                sat_probe(sat, h[0][0]);
#pragma unroll
                for (int m = 0; m < 2; ++m) { h[m] = elu4(h[m]); t[m] = ld4(W + wf_.n_bq1 + 16 * m + 4 * g); }
                split8(h[0], h[1], xh, xl);
                mfma16_multi<2>(Lq1, 0, 1, 0, lane, xh, xl, t);
                sat_probe(sat, t[0][0]);
#pragma unroll
                for (int m = 0; m < 2; ++m) { t[m] = elu4(t[m]); h[m] = ld4(W + wf_.n_bq2 + 16 * m + 4 * g); }
                split8(t[0], t[1], xh, xl);
                mfma16_multi<2>(Lq2, 0, 1, 0, lane, xh, xl, h);
#pragma unroll
                for (int m = 0; m < 2; ++m) st[m] += h[m];
            } else {           // ppm: 64 -> 32, no bias, xyz component role - 1             (:148, :152)
                f32x4 a[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
                const float* zc = zr + 64 + (role - 1) * 64;
#pragma unroll
                for (int kgp = 0; kgp < 2; ++kgp) {
                    split8(ld4(zc + 32 * kgp + 4 * g), ld4(zc + 32 * kgp + 16 + 4 * g), xh, xl);
                    mfma16_multi<2>(Lpp, 0, 2, kgp, lane, xh, xl, a);
                }
#pragma unroll
                for (int m = 0; m < 2; ++m) st[m] += a[m];
            }
            sat_probe(sat, st[0][0]);
            if (valid) mag_flush_at(st[0], st[1], state_limit_of(flags), flags, i);
            if (i == 0) { st[0] = f32x4{0, 0, 0, 0}; st[1] = st[0]; }                               // :239-240 sink
            if (valid) {
                float* dst = role == 0 ? q_state + (size_t)i * S : p_state + (size_t)i * 96 + (role - 1) * 32;
                st4(dst + 4 * g, st[0]); st4(dst + 16 + 4 * g, st[1]);
            }
        }
        if (!do_prep) { if (valid) sat_flush_at(sat, flags, i); sat = 0.0f; continue; }
        // exchange the tile state between the four roles
        st4(xs + (2 * role) * 256 + lane * 4, st[0]);
        st4(xs + (2 * role + 1) * 256 + lane * 4, st[1]);
        __syncthreads();
        f32x4 q[2], p[3][2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            q[m] = ld4(xs + m * 256 + lane * 4);
#pragma unroll
            for (int c = 0; c < 3; ++c) p[c][m] = ld4(xs + (2 + 2 * c + m) * 256 + lane * 4);
        }
        __syncthreads();                                     // the next iteration overwrites the exchange buffer

        const float* Lua = wl_ + NL_UA, *Lgc = wl_ + NL_GC, *Lnq = wl_ + NL_NQ;
        f32x4 pn[2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                pn[m][r] = norm3_fast(p[0][m][r], p[1][m][r], p[2][m][r]);
        f16x8 xnh[2], xnl[2], ph[3], pl[3];
        split8(q[0], q[1], xnh[0], xnl[0]);
        split8(pn[0], pn[1], xnh[1], xnl[1]);
#pragma unroll
        for (int c = 0; c < 3; ++c) split8(p[c][0], p[c][1], ph[c], pl[c]);

        float* cen = rec_cen + (size_t)i * REC_CEN;
        float* nb = rec_nb + (size_t)i * REC_A;
        {   // [U | A] output blocks 4 role .. 4 role + 3 (U = blocks 0..7 carries b1, A = blocks 8..15)
            const int ob = 4 * role;
            f32x4 a[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] = ob < 8 ? ld4(W + wp_.n_b1s + 16 * (ob + j) + 4 * g) : f32x4{0, 0, 0, 0};
#pragma unroll
            for (int kgp = 0; kgp < 2; ++kgp) mfma16_multi<4>(Lua, ob, 2, kgp, lane, xnh[kgp], xnl[kgp], a);
            sat_probe(sat, a[0][0]);
            if (valid) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (ob < 8) st4_finite(cen + (ob + j) * 64 + 3 * 16 + 4 * g, a[j]);
                    else st4(nb + (ob + j - 8) * 16 + 4 * g, a[j]);                   // A_j[16 fb + 4g + r]
                }
            }
        }
        {   // G blocks 2 role, 2 role + 1
            const int ob = 2 * role;
            f32x4 a[2][3];
            f16x8 wh[2], wl[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float* fr = Lgc + (size_t)((ob + j) * 2) * 256 + lane * 4;
                wh[j] = ld8h(fr); wl[j] = PESTO_WL(fr);
#pragma unroll
                for (int c = 0; c < 3; ++c) a[j][c] = f32x4{0, 0, 0, 0};
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) { a[0][c] = MFMA16(wh[0], ph[c], a[0][c]); a[1][c] = MFMA16(wh[1], ph[c], a[1][c]); }
#pragma unroll
            for (int c = 0; c < 3; ++c) { a[0][c] = MFMA16(wh[0], pl[c], a[0][c]); a[1][c] = MFMA16(wh[1], pl[c], a[1][c]); }
#pragma unroll
            for (int c = 0; c < 3; ++c) { a[0][c] = MFMA16(wl[0], ph[c], a[0][c]); a[1][c] = MFMA16(wl[1], ph[c], a[1][c]); }
#pragma unroll
            for (int c = 0; c < 3; ++c) sat_probe(sat, a[0][c][0]);
            if (valid) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int c = 0; c < 3; ++c) st4_finite(cen + (ob + j) * 64 + c * 16 + 4 * g, a[j][c]);
            }
        }
        if (role == 3) {   // node queries Q = nqm(X_n): 64 -> 32 -> 32 -> 12                        (:119)
            f32x4 h[2], t[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) { h[m] = ld4(W + wp_.n_bn0 + 16 * m + 4 * g); t[m] = ld4(W + wp_.n_bn1 + 16 * m + 4 * g); }
#pragma unroll
            for (int kgp = 0; kgp < 2; ++kgp) mfma16_multi<2>(Lnq, 0, 2, kgp, lane, xnh[kgp], xnl[kgp], h);
            sat_probe(sat, h[0][0]);
            split8(elu4(h[0]), elu4(h[1]), xh, xl);
            mfma16_multi<2>(Lnq + 2048, 0, 1, 0, lane, xh, xl, t);
            sat_probe(sat, t[0][0]);
            f32x4 qq[1] = {ld4(W + wp_.n_bn2s + 4 * g)};
            split8(elu4(t[0]), elu4(t[1]), xh, xl);
            mfma16_multi<1>(Lnq + 3072, 0, 1, 0, lane, xh, xl, qq);
            sat_probe(sat, qq[0][0]);
            if (valid) st4(cen + 512 + 4 * g, qq[0]);
        }
        if (valid) sat_flush_at(sat, flags, i);      // (this lane's probes cover MFMA column e = atom i of the tile)
        sat = 0.0f;
    }
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
// =============================================================================================== edge kernel
#if defined(PESTO_TRACE32)      // developer build: timeline of ONE wave (block 0, wave 0), no per-phase accumulators (they cost registers)
__device__ unsigned long long g_trace32[4096];     // four timelines of 1024 entries: the last nn = 8 / 16 / 32 / 64 launch
__device__ int g_trace32_base;                      // (set by the kernel's first mark, id >= 60 + log2(nn / 8))
__device__ __forceinline__ void trace32(int& n, int id) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && n < 1023) {
        if (id >= 60 && id < 64) { n = 0; g_trace32_base = (id - 60) * 1024; }
        g_trace32[g_trace32_base + n++] = ((unsigned long long)id << 56) | (__builtin_readcyclecounter() & 0xffffffffffffffull);
        g_trace32[g_trace32_base + n] = 0;
    }
}
#define PHASE_MARK(k) trace32(tr_n, 30 + (k))
#define PHASE_INIT() do {} while (0)
#define PHASE_DECL() do {} while (0)
#define PHASE_FLUSH() do {} while (0)
#define TRACE32(id) trace32(tr_n, id)
#define P32_MARK(k) trace32(tr_n, k)
#elif defined(PESTO_PROFILE_PHASES)   // developer build: per-phase wave cycles (s_memtime), printed by pesto_destroy
__device__ unsigned long long g_phase_cycles[12];
#define PHASE_MARK(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); phase_acc_[k] += now_ - tmark_; tmark_ = now_; } while (0)
#define PHASE_INIT() tmark_ = __builtin_readcyclecounter()
#define PHASE_DECL() unsigned long long tmark_ = 0, phase_acc_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define PHASE_FLUSH() do { if ((threadIdx.x & 63) == 0) for (int k_ = 0; k_ < 12; ++k_) atomicAdd(&g_phase_cycles[k_], phase_acc_[k_]); } while (0)
#define TRACE32(id) do {} while (0)
#define P32_MARK(k) do {} while (0)
#else
#define PHASE_MARK(k) do {} while (0)
#define PHASE_INIT() do {} while (0)
#define PHASE_DECL() do {} while (0)
#define PHASE_FLUSH() do {} while (0)
#define TRACE32(id) do {} while (0)
#define P32_MARK(k) do {} while (0)
#endif
// ROWS = edge rows of a work item: 64, or 16 for the one-tile items of the sixteen-wave node-wave workgroups (twelve item waves: LDS)
template <int ROWS>
struct alignas(16) EdgeWaveScratchT {      // 16-byte multiple: the rows are read / written as float4 (ds_read / ds_write_b128)
    int nb[ROWS];          // neighbour id per row
    float geo[5][ROWS];    // r_hat x, y, z, d per row (SoA); row 4 = 1.0 (k = 3 slot of the centre MFMA's B operand)
    float wts[8][ROWS];    // attention weights [h*4 + part][row]: part 0 scalar, 1..3 the vector chunks
    float wsum[8][2];    // per centre: sum over edges of the part-2 weights (multiplies p_i)
    float z3buf[2][2][96];  // [centre sel][h][c*32+s]: sum_e w3[h][e] p_j(e), staged for the final combine
};
using EdgeWaveScratch = EdgeWaveScratchT<64>;
// per-wave scratch of the 32-edge-tile kernel (M32, further down)
struct alignas(16) EdgeWaveScratch32 {
    int nb[64];               // neighbour id per row of the work item
    float geo[4][64];         // r_hat x, y, z, d per row
    float stage[32 * 36];     // 32 edges x 9 units of 16 B: the p_j . r operands, then the A_j chunks of a tile (aliased in time);
                              // at the end of a centre the first 384 floats hold the p_j sums [slot][h][96]
    float wt[10][32];         // unnormalised attention weights of the current tile: rows h: scalar; 2 + 3 h + c: part 1 x r_c; 8 + h: part 3
    float stat[2][2][4];      // [centre slot][h]: 1 / s_scalar, 1 / s_vector, (sum of the part-2 weights) / s_vector
};
constexpr int ST32 = 36;      // floats per edge row of the staging area (9 units: an odd stride spreads 16 consecutive rows over all banks)
// offsets of the NEXT layer's prepare tables (the [U|A], G and nqm fragments / biases of LayerW): the finishing waves of the edge
// kernel write that layer's centre / neighbour records right behind the state update (k_node16's prepare half, same arithmetic)
struct PrepW { int32_t h_ua, h_gc, h_n0, n_b1s, n_bn0, n_bn1, n_bn2; };
// XCH_FLOATS: tile-state exchange of the prepare phase (FIN kernels): per 16-centre tile [q0 q1 p00 p01 p10 p11 p20 p21][fg 4][column][4];
// the second tile of a twelve-wave workgroup holds 8 centres (24 per iteration) and is stored compactly: 2048 + 1024 floats
constexpr int XCH_FLOATS = 3072;
// NE = waves that process work items. NE == WPB: every wave does, the finish / prepare phase runs behind workgroup rendezvous.
// NE < WPB ("node waves"): the other WPB - NE waves ONLY finish / prepare, fed through LDS queues without any workgroup barrier:
// two generations of staged Z rows per edge wave and of the 16-centre state exchange.
constexpr int XF_POST = 0, XF_READY = 2, XF_CONSUMED = 4, XF_READY2 = 6;      // xflag slots: slices posted per tile | rows staged per generation | generations read
template <int WPB, bool HY, bool XCH = false, int NE = WPB, bool M32 = false, int ROWS = 64>
struct EdgeSmem {
    static constexpr int GEN = (XCH && NE < WPB) ? 2 : 1;
    float w[M32 ? EDGE_LDS_FLOATS_32 : HY ? EDGE_LDS_FLOATS_HY : EDGE_LDS_FLOATS];
    typename std::conditional<M32, EdgeWaveScratch32, EdgeWaveScratchT<ROWS>>::type ws[NE];
    float zrows[NE][GEN][2][256];   // Zq | Zp staging per centre: two rows per edge wave (and generation)
    float xch[XCH ? (NE < WPB ? (WPB - NE) / 4 * 2 * 2048 : XCH_FLOATS) : 4];      // (node waves: two generations per team)
    // bias of the value network's last layer, four copies per feature: the accumulator tile of feature column e starts as (b, b, b, b) -
    // one ds_read_b128 instead of a 4-byte read + four v_mov per block (16 v_mov per tile). Filled by the kernel's prologue.
    alignas(16) float b3v4[(HY && !M32) ? 256 : 4];
    int xflag[8];        // monotone counters (see XF_*)
};
// poll an LDS counter of this workgroup (all its waves are resident); SLEEP x 64 cycles between two looks: a polling wave takes issue
// slots from the waves of its SIMD, so a long expected wait polls rarely
template <int SLEEP = 1>
__device__ __forceinline__ void lds_wait_ge(int* flag, int target) {
    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < target)
        __builtin_amdgcn_s_sleep(SLEEP);
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void lds_signal(int* flag, bool one_lane) {     // count one event behind this wave's LDS traffic
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (one_lane) __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Operands of the first edge layer for feature block fb of one 16-edge tile, fetched one or two blocks AHEAD of
// their use (explicit software prefetch: with 2 waves per SIMD the gather latency is not hidden otherwise).
struct L1Ops { f32x4 a4, c0, c1, c2; float cenA, cenB; };

template <int NN>
__device__ __forceinline__ L1Ops l1_fetch(int fb, int lane, int g, const float* __restrict__ cenA, const float* __restrict__ cenB,
                                          const float* __restrict__ recj) {
    L1Ops o;
    const float* rp = recj + (fb * 4 + g) * 16;
#ifdef PESTO_ABL_L1QUARTER   // ablation: a quarter of the neighbour-record gather traffic, same arithmetic (results wrong)
    o.a4 = ld4(rp); o.c0 = o.a4; o.c1 = o.a4; o.c2 = o.a4;
#else
    o.a4 = ld4(rp); o.c0 = ld4(rp + 4); o.c1 = ld4(rp + 8); o.c2 = ld4(rp + 12);
#endif
    o.cenA = cenA[fb * 64 + lane];
    o.cenB = NN == 8 ? cenB[fb * 64 + lane] : 0.0f;
    return o;
}

// h1[r] = ELU(pre-activation of feature 16fb+4g+r, edge e)
template <int NN>
__device__ __forceinline__ f32x4 l1_compute(const L1Ops& o, int fb, int g, float bgA, float bgB, const float* __restrict__ wd,
                                            float d, float rx, float ry, float rz) {
    f32x4 acc = MFMA(o.cenA, bgA, (f32x4{0, 0, 0, 0}));          // sum_c G_i[c] r_c + U_i, centre A columns
    if (NN == 8) acc = MFMA(o.cenB, bgB, acc);                   // second centre of the tile
    const f32x4 w4 = ld4(wd + 16 * fb + 4 * g);
    f32x4 h;
#pragma unroll
    for (int r = 0; r < 4; ++r) h[r] = elu_f(acc[r] + o.a4[r] + d * w4[r] + rx * o.c0[r] + ry * o.c1[r] + rz * o.c2[r]);
    return h;
}

// per-tile addressing: centre record(s), neighbour record of this lane's edge, geometry
struct TileCtx { const float *cenA, *cenB, *recj, *recj_p; float rx, ry, rz, d, bgA, bgB; };

template <int NN, bool HY = false, class WS = EdgeWaveScratch>
__device__ __forceinline__ TileCtx tile_ctx(int t, int e, int g, int c0, int N1, const WS& ws,
                                            const float* __restrict__ rec_nb, const float* __restrict__ rec_cen) {
    TileCtx c;
    const int row = 16 * t + e;
    const int aA = NN == 8 ? 2 * t : (16 * t) / NN;
    c.cenA = rec_cen + (size_t)ABL_CEN(min(c0 + aA, N1 - 1)) * REC_CEN;
    c.cenB = rec_cen + (size_t)ABL_CEN(min(c0 + aA + 1, N1 - 1)) * REC_CEN;
    c.rx = ws.geo[0][row]; c.ry = ws.geo[1][row]; c.rz = ws.geo[2][row]; c.d = ws.geo[3][row];
    const float bg = ws.geo[g == 3 ? 4 : g][row];      // B operand of the centre MFMA: (r_x, r_y, r_z, 1)[k = g], one LDS read
    c.bgA = (NN == 8 && e >= 8) ? 0.0f : bg;
    c.bgB = (NN == 8 && e >= 8) ? bg : 0.0f;
    c.recj = rec_nb + (size_t)ABL_NB(ws.nb[row]) * (HY ? REC_A : REC_NB);
    c.recj_p = rec_nb + (size_t)ABL_NB(ws.nb[16 * t + ((16 * g + e) >> 2)]) * (HY ? REC_A : REC_NB);
    return c;
}

// ---- hybrid first layer: neighbour terms = A_j (gathered, 512 B) + W[:,161:193] (p_j . r) on the matrix cores
// B operand of the W1P MFMAs for one tile: lane (edge e, kg = g) holds p_j(e) . r_hat for s = 8g .. 8g+7, as f16 hi/lo
// Gathers are issued in a PRODUCER lane layout, lane = 4 * edge + chunk: the four lanes of an edge read 64 contiguous bytes, so
// a quarter-wave (what the vector L1 processes per pass) touches 4 cache lines instead of 16 - the L1's line-request rate,
// not bytes or VALU, bounded this kernel. The MFMA operand layout wants lane = 16 * chunk + edge; values move there with
// ds_bpermute (LDS crossbar, no LDS storage): MFMA lane (e, g) pulls from producer lane 4e + g.
__device__ __forceinline__ float bperm(int src_byte, float v) {
#ifdef PESTO_ABL_NOBPERM   // ablation: the A_j chunks stay in the producer lanes (results wrong): -16 ds_bpermute per tile and pass
    return v;
#endif
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_byte, __builtin_bit_cast(int, v)));
}
// Producer lanes hold the 16-byte pieces of an edge in swizzled order - lane 4 e + s holds piece s ^ (e >= 8 ? 2 : 0) - so that the 32
// consumer lanes ds_bpermute serves together pull from 32 different banks (lanes l and l + 32 share one; unswizzled, edges e and e + 8
// collided: 10 of the kernel's 14.8 % SQ_LDS_BANK_CONFLICT, profiles/r03_lds_conflict_ablation.txt). Pure data movement: same bits.
__device__ __forceinline__ int prod_piece(int lane) { return (lane & 3) ^ ((lane >> 5) << 1); }                    // piece a producer lane loads
__device__ __forceinline__ int cons_src(int lane) { return (4 * (lane & 15) + ((lane >> 4) ^ (((lane >> 3) & 1) << 1))) << 2; }   // byte address of the lane a consumer pulls
__device__ __forceinline__ f32x4 to_mfma_lanes(f32x4 v, int lane) {
    const int src = cons_src(lane);
    return f32x4{bperm(src, v[0]), bperm(src, v[1]), bperm(src, v[2]), bperm(src, v[3])};
}
// First layer of the four feature blocks fb0 .. fb0+3 of tile t, in three steps so that the caller can software-pipeline:
//   l1_issue : ALL global loads of the tile (p_j rows, A_j chunks, centre record columns) issued together - left to itself the
//              scheduler issued the A_j / centre loads one block at a time, each followed by a full vmcnt(0) wait (five
//              serialized memory round trips per tile);
//   l1_head  : p_j . r_hat -> f16 hi/lo -> MFMA lane layout, A_j chunks -> MFMA lane layout, centre MFMAs (consumes the raw loads);
//   l1_tail  : the W1P MFMAs, distance term, ELU.
struct L1Raw { f32x4 x0, x1, y0, y1, z0, z1, a4[4]; float cA[4], cB[4]; };
struct L1Head { f16x8 fh, fl; f32x4 acc[4]; float d; };

template <int NN, class WS = EdgeWaveScratch>
__device__ __forceinline__ L1Raw l1_issue(int fb0, int t, int lane, const TileCtx& tc, const WS& ws,
                                          const float* __restrict__ p_state) {
    L1Raw r;
    const int rp = 16 * t + (lane >> 2);               // producer lane: edge rp, piece prod_piece(lane) (32 bytes of p_j, 16 of A_j)
    const int pc = prod_piece(lane);
    const float* pj = p_state + (size_t)ABL_NB(ws.nb[rp]) * 96 + 8 * pc;
    r.x0 = ld4(pj); r.x1 = ld4(pj + 4); r.y0 = ld4(pj + 32); r.y1 = ld4(pj + 36); r.z0 = ld4(pj + 64); r.z1 = ld4(pj + 68);
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) {
        r.a4[fb] = ld4(tc.recj_p + (fb0 + fb) * 16 + 4 * pc);
        r.cA[fb] = tc.cenA[(fb0 + fb) * 64 + lane];
        r.cB[fb] = NN == 8 ? tc.cenB[(fb0 + fb) * 64 + lane] : 0.0f;
    }
    return r;
}

template <int NN, class WS = EdgeWaveScratch>
__device__ __forceinline__ L1Head l1_head(const L1Raw& r, int t, int lane, const TileCtx& tc, const WS& ws) {
    L1Head o;
    const int rp = 16 * t + (lane >> 2);
    // p_j(ep) . r_hat(ep) for s = 8 chunk .. 8 chunk + 7 (model_operations.py:115), split, moved to the MFMA lane layout
    const float rx = ws.geo[0][rp], ry = ws.geo[1][rp], rz = ws.geo[2][rp];
    const f32x4 a = r.x0 * rx + r.y0 * ry + r.z0 * rz;
    const f32x4 b = r.x1 * rx + r.y1 * ry + r.z1 * rz;
    f16x8 fh, fl;
    split8(a, b, fh, fl);
    const int src = cons_src(lane);
    u32x4 hp = __builtin_bit_cast(u32x4, fh), lp = __builtin_bit_cast(u32x4, fl);
#ifndef PESTO_ABL_NOBPERM2   // ablation: the p_j . r operands stay in the producer lanes too (results wrong): -8 ds_bpermute per tile and pass
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        hp[j] = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)hp[j]);
        lp[j] = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)lp[j]);
    }
#endif
    o.fh = __builtin_bit_cast(f16x8, hp);
    o.fl = __builtin_bit_cast(f16x8, lp);
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) {
        // A_j + sum_c G_i[c] r_c + U_i: the gathered neighbour term is the accumulator the centre MFMA starts from (no separate add)
        o.acc[fb] = MFMA(r.cA[fb], tc.bgA, to_mfma_lanes(r.a4[fb], lane));
        if (NN == 8) o.acc[fb] = MFMA(r.cB[fb], tc.bgB, o.acc[fb]);
    }
    o.d = tc.d;
    return o;
}

// One-tile work items (ONEP): the p_j . r_hat operand of a tile is the same in both passes - the second pass reuses the (hi, lo) pair of
// the first and only fetches what differs per feature-block half: the A_j chunks and the centre-record columns of blocks fb0 .. fb0 + 3.
// Issued right behind the first pass's head, they land under the key networks and the softmax (a one-tile item has no next tile whose
// loads it could overlap with: each of its dependent round trips is paid in full). Same values, same order: same bits as two passes.
struct L1RawAC { f32x4 a4[4]; float cA[4], cB[4]; };
template <int NN>
__device__ __forceinline__ L1RawAC l1_issue_ac(int fb0, int lane, const TileCtx& tc) {
    L1RawAC r;
    const int pc = prod_piece(lane);
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) {
        r.a4[fb] = ld4(tc.recj_p + (fb0 + fb) * 16 + 4 * pc);
        r.cA[fb] = tc.cenA[(fb0 + fb) * 64 + lane];
        r.cB[fb] = NN == 8 ? tc.cenB[(fb0 + fb) * 64 + lane] : 0.0f;
    }
    return r;
}
template <int NN>
__device__ __forceinline__ L1Head l1_head_ac(const L1RawAC& r, f16x8 fh, f16x8 fl, int lane, const TileCtx& tc) {
    L1Head o;
    o.fh = fh; o.fl = fl;
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) {
        o.acc[fb] = MFMA(r.cA[fb], tc.bgA, to_mfma_lanes(r.a4[fb], lane));
        if (NN == 8) o.acc[fb] = MFMA(r.cB[fb], tc.bgB, o.acc[fb]);
    }
    o.d = tc.d;
    return o;
}

__device__ __forceinline__ void l1_tail(L1Head& o, int fb0, int lane, int g, const float* __restrict__ w1p, const float* __restrict__ wd,
                                        f32x4* h1, float& sat) {
#pragma unroll
    for (int m0 = 0; m0 < 4; m0 += 2) {
        f16x8 wh[2], wl[2];
#pragma unroll
        for (int ml = 0; ml < 2; ++ml) {
            const float* fr = w1p + (size_t)((fb0 + m0 + ml) * 2) * 256 + lane * 4;
            wh[ml] = ld8h(fr); wl[ml] = PESTO_WL(fr);
        }
#pragma unroll
        for (int ml = 0; ml < 2; ++ml) o.acc[m0 + ml] = MFMA16(wh[ml], o.fh, o.acc[m0 + ml]);
#pragma unroll
        for (int ml = 0; ml < 2; ++ml) o.acc[m0 + ml] = MFMA16(wh[ml], o.fl, o.acc[m0 + ml]);
#pragma unroll
        for (int ml = 0; ml < 2; ++ml) o.acc[m0 + ml] = MFMA16(wl[ml], o.fh, o.acc[m0 + ml]);
    }
    sat_probe(sat, o.acc[0][0]);      // p_j . r_hat beyond the f16 range
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) {
        const f32x4 w4 = ld4(wd + 16 * (fb0 + fb) + 4 * g);
        h1[fb] = elu4s(o.acc[fb] + o.d * w4);      // every term arrives in the log2 domain
    }
}

// =============================================================================================== 32-edge tiles (M32)
// The same layer on v_mfma_f32_32x32x16_f16: one MFMA covers 32 edges x 32 features. Why (profiles/microbench/r03_shadow.txt): next
// to the 4-pass 16x16x32 instruction almost nothing else issues (its cost and the cost of every VALU instruction simply add up),
// while the 8-pass 32x32x16 one takes the same time per FLOP and hides ~5 VALU instructions each; a weight fragment read from LDS
// feeds twice as many edges. What the wider tile changes:
//   * ONE pass per tile: keys, logits, a tile-local softmax (unnormalised weights e = exp(l - m); for nn = 64 the running maximum of
//     the centre's two tiles with the usual rescaling of the partial sums), values, weighted sums; the normalisation 1 / sum is applied
//     once per centre at the end. The gathers (A_j, p_j) and p_j . r are done once per edge, not once per pass.
//   * Gathers stay in a producer lane layout (consecutive lanes read consecutive 16-byte pieces of one atom's record, few cache lines
//     per quarter-wave) and reach the MFMA operand layout (lane = edge) through a 4.5 KB LDS staging area per wave (ds_write_b128 /
//     ds_read_b128, rows of 9 x 16 B: conflict-free both ways) - ds_bpermute cannot fill a 32-edge operand from 16-edge producer rounds.
//   * A_j is read from the staging area straight into the accumulators (no add); the centre terms U_i + sum_c r_c G_i[c] AND the distance
//     term w_d d are ONE extra K = 16 MFMA per 32 features: A = (G0 G1 G2 U)(hi) (G0 G1 G2 U)(lo) | (G0 G1 G2 U)(hi) (wd_hi wd_lo wd_hi 0),
//     B = (rx ry rz 1)(hi) x 2 | (rx ry rz 0)(lo) (d_hi d_hi d_lo 0) - the three products of the hi/lo split.
// D tile: lane (n = l & 31, hl = l >> 5), register v <-> row 8 (v >> 2) + 4 hl + (v & 3), column n. Registers [8 ks .. 8 ks + 7] of a D
// tile are the B (or A) operand of k-step ks of the next layer; the weight fragments are laid out to match (pesto_schema.h, EL32_*).
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)


__device__ __forceinline__ f32x4 bufld4(__amdgpu_buffer_rsrc_t r, int byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}
__device__ __forceinline__ float bufld1(__amdgpu_buffer_rsrc_t r, int byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}
__device__ __forceinline__ float bufld1s(__amdgpu_buffer_rsrc_t r, int byte_off, int s_off) {      // s_off: wave-uniform (SGPR) part
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, s_off, 0));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, 0xfffffffc, 0x00020000);
}
// op(x[l], x[l ^ 16]) / op(x[l], x[l ^ 32]) on the VALU (gfx950 lane swaps; both operands are the same value)
template <int CTRL>
__device__ __forceinline__ float dpp_bc(float x) {      // DPP source with bound_ctrl (no `old` value: the compiler folds it into the consumer)
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
// v_permlane16_swap / v_permlane32_swap exchange rows (halves) BETWEEN two registers; with a copy of x in the second register the two
// results are x and x[l ^ 16] (x[l ^ 32]) in some order, so a symmetric op needs no select. Inline asm (validated in
// profiles/microbench/permlane_test.hip): the compiler's builtin returned the same register for both results here. Only called on values
// produced by VALU code after this wave's last MFMA result has been read (no matrix instruction of the wave is in flight).
template <bool IS_MAX>
__device__ __forceinline__ float xrow(float x) {
    float a = x, b;
    asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "=&v"(b));
    return IS_MAX ? fmaxf(a, b) : a + b;
}
template <bool IS_MAX>
__device__ __forceinline__ float xhalf(float x) {
    float a = x, b;
    asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "=&v"(b));
    return IS_MAX ? fmaxf(a, b) : a + b;
}
// Reduce-scatter steps of the centre epilogue (EPI2): a and b are two accumulators that BOTH need the sum over a pair of lane rows
// (16-lane rows r, r ^ 1) resp. lane halves. v_permlane16_swap exchanges the odd rows of a with the even rows of b - afterwards a + b is,
// in the even rows, a's total over the row pair and, in the odd rows, b's: one swap + one add for two values (the ds_bpermute form:
// two address computations, two permutes, two adds - and every lane ends with both totals although only one row of lanes stores them).
// Same operands, commutative add: the same bits as x += shfl_xor(x, 16). Inline asm as xrow above (the builtin is unusable here);
// the callers fence the block with sched_barrier so that no matrix instruction is in flight around it.
__device__ __forceinline__ float swap_add_rows(float a, float b) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float swap_add_halves(float a, float b) {      // lanes 0..31: a's total over (l, l + 32); lanes 32..63: b's
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}
// reduction over the lanes of one centre inside a 32-lane half (16 lanes for nn = 16, the whole half otherwise)
template <int NN, bool IS_MAX>
__device__ __forceinline__ float centre_reduce(float x) {
    x = row_reduce<true, IS_MAX>(x);
    if (NN >= 32) x = xrow<IS_MAX>(x);
    return x;
}
__device__ __forceinline__ unsigned pack_h2(float lo, float hi) {      // (f16(lo) | f16(hi) << 16), round to nearest
    const f32x2 x = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(x, f16x2));
}
// x -> packed (hi pair, lo pair) of two values: hi = rn16(x), lo = rn16(x - hi)
__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) {
    const f32x2 x = {a, b};
    const f16x2 h = __builtin_convertvector(x, f16x2);
    float m1 = -1.0f;
    asm("" : "+s"(m1));
    f16x2 l;
    l[0] = (_Float16)__builtin_fmaf((float)h[0], m1, a);
    l[1] = (_Float16)__builtin_fmaf((float)h[1], m1, b);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ f16x8 frag32(const float* tab, int idx, int lane) { return ld8h(tab + idx * 256 + lane * 4); }

#ifdef PESTO_DEBUG32     // developer build: intermediates of the work item that holds centre g_dbg32_centre
__device__ float g_dbg32[4096];
__device__ int g_dbg32_centre = 5;
#define DBG32(idx, val) do { if (dbg_) g_dbg32[(idx)] = (val); } while (0)
#else
#define DBG32(idx, val) do {} while (0)
#endif
// One work item (TI/2 tiles of 32 edges = A whole centres) of the M32 kernel. Leaves the complete Z rows of its centres in zrow[slot].
template <int NN, int T32>
__device__ __forceinline__ void edge_item32(EdgeWaveScratch32& ws, const float* __restrict__ smw, float (*zrow)[256], int sub, int lane_in,
                                            int c0, int N1, __amdgpu_buffer_rsrc_t r_nb, __amdgpu_buffer_rsrc_t r_cen, __amdgpu_buffer_rsrc_t r_p,
                                            float inv_sdk, float& sat, int& tr_n) {
    constexpr int CPT = NN == 16 ? 2 : 1;            // centres per tile
    constexpr int NCS = NN == 16 ? 2 : 1;            // accumulator sets per lane (centres whose edges one lane's V registers cover)
    constexpr float L2E = 1.44269504088896340736f;
    // sum / max over the lanes of one centre inside a 32-lane half: four DPP steps inside the arithmetic instruction (v_add_f32_dpp; the max
    // as inline v_max_f32_dpp - the compiler's fmaxf canonicalises both operands first: three instructions per step) + one row swap
    auto csum = [](float x) {
        x += dpp_bc<0xB1>(x); x += dpp_bc<0x4E>(x); x += dpp_bc<0x141>(x); x += dpp_bc<0x140>(x);
        if (NN >= 32) x = xrow<false>(x);
        return x;
    };
    auto cmax = [](float x) {
        asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1" : "+v"(x));
        if (NN >= 32) x = xrow<true>(x);
        return x;
    };
#ifdef PESTO_DEBUG32
    const bool dbg_ = g_dbg32_centre >= c0 && g_dbg32_centre < c0 + 16 * 2 * T32 / NN;
#endif
    // running state of a centre that spans two tiles (nn = 64): maxima, sums, partial Z (rescaled when the maximum moves)
    float run_ms[2] = {0.f, 0.f}, run_mv[2] = {0.f, 0.f};
    float S_s[2] = {0.f, 0.f}, S_v[2] = {0.f, 0.f}, W2[2] = {0.f, 0.f};
    float zq[NCS][2], zp1[NCS][2][3];
    f32x4 z3[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
#pragma unroll
    for (int cs = 0; cs < NCS; ++cs)
#pragma unroll
        for (int h = 0; h < 2; ++h) { zq[cs][h] = 0.f; zp1[cs][h][0] = zp1[cs][h][1] = zp1[cs][h][2] = 0.f; }
    float p_own[NCS][3];                      // the centre's own p (second block of Vp, model_operations.py:133), feature n
    // (prefetching the next tile's p_j rows into registers - all 48 behind the first value layer, all 48 or only the first round of 24
    // at the end of the tile - makes the register allocator spill in the hot loop: nn = 64 275 -> 330-340 us per launch,
    // profiles/r03_m32_ab.txt; the loads are therefore issued at the top of their own tile)

#pragma unroll 1
    for (int t = 0; t < T32; ++t) {
        const int row0 = 32 * t;
        // (the lane-derived addresses of the gathers are re-derived per tile from an opaque copy of the lane index: hoisted out of
        // the tile loop as loop invariants they occupy thirty registers for the whole item)
        int lane_o = lane_in;
        asm volatile("" : "+v"(lane_o));
        const int lane = lane_o, n = lane & 31, hl = lane >> 5, n_o = n, hl_o = hl;
        const int esub = lane / 24, quad = lane - 24 * esub, es = esub == 2 ? 0 : esub;     // p_j sums: lane = (edge half, 16-byte piece of the 96-vector)
        // centre of column n of this tile
        const int cen_i = min(c0 + (NN == 64 ? 0 : NN == 32 ? t : (n_o >> 4)), N1 - 1);
        const int cen_off = cen_i * (REC_CEN * 4);
        // ------------------------------------------------------------------ gathers of the tile
        int aj_off[4];       // A_j: lane = 8 * edge + 16-byte piece of the 128-byte chunk of one 32-feature block; 8 edges per load
#pragma unroll
        for (int i = 0; i < 4; ++i) aj_off[i] = ws.nb[row0 + 8 * i + (lane_o >> 3)] * (REC_A * 4) + 16 * (lane_o & 7);
        f32x4 aj[2][4];      // two chunks in flight (blocks 0, 1 now; 2, 3 during the key networks)
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) aj[k][i] = bufld4(r_nb, aj_off[i] + 128 * k);
        // centre records: (G0, G1, G2, U) of feature 32 rb + n for every centre of the tile (rows of the centre MFMA's A operand):
        // one lane-dependent address, the block / piece as immediate offsets, the centre as the wave-uniform offset
        float cg[CPT][4][4];
        const int cg_lane = ((n_o >> 4) * 64 + (n_o & 15)) * 4;
#pragma unroll
        for (int cc = 0; cc < CPT; ++cc) {
            const int co = __builtin_amdgcn_readfirstlane(min(c0 + (NN == 64 ? 0 : NN == 32 ? t : cc), N1 - 1) * (REC_CEN * 4));
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int kg = 0; kg < 4; ++kg) cg[cc][rb][kg] = bufld1s(r_cen, cg_lane + (2 * rb * 64 + kg * 16) * 4, co);
        }
        float QA[6], QV[6];                   // QA: queries of this lane's first key group (scalar for hl = 0, vector for hl = 1); QV: vector
#pragma unroll
        for (int j = 0; j < 6; ++j) { QA[j] = bufld1(r_cen, cen_off + (512 + 6 * hl_o + j) * 4); QV[j] = bufld1(r_cen, cen_off + (518 + j) * 4); }
        if (t % (NN == 64 ? 2 : 1) == 0) {
#pragma unroll
            for (int cs = 0; cs < NCS; ++cs) {
                const int ci = min(c0 + (NN == 32 ? t : cs), N1 - 1);
#pragma unroll
                for (int c = 0; c < 3; ++c) p_own[cs][c] = bufld1(r_p, ci * 384 + (c * 32 + n_o) * 4);
            }
        }
        P32_MARK(0);
        // ------------------------------------------------------------------ geometry operand of the centre MFMA
        const float gx = ws.geo[0][row0 + n], gy = ws.geo[1][row0 + n], gz = ws.geo[2][row0 + n], gd = ws.geo[3][row0 + n];
        u32x4 bcen[CPT];
        {
            unsigned hxy, lxy, hz1, lz0, hd, ld;
            split2(gx, gy, hxy, lxy);
            split2(gz, gd, hz1, lz0);           // (rz_hi | d_hi), (rz_lo | d_lo)
            hd = hz1 >> 16; ld = lz0 >> 16;
            const unsigned one = 0x3c00u;       // 1.0 in f16
            const unsigned rz1 = (hz1 & 0xffffu) | (one << 16);
            u32x4 b;
            if (hl == 0) b = u32x4{hxy, rz1, hxy, rz1};
            else b = u32x4{lxy, lz0 & 0xffffu, hd | (hd << 16), ld};
            if (NN == 16) {      // two centres per tile: each centre's MFMA sees only its own 16 columns
                const u32x4 zero = u32x4{0, 0, 0, 0};
                bcen[0] = n < 16 ? b : zero; bcen[CPT - 1] = n < 16 ? zero : b;
            } else {
                bcen[0] = b;
            }
        }
        // ------------------------------------------------------------------ p_j . r_hat (model_operations.py:115) -> f16 hi/lo -> staging
        {
            f32x4 pj[2][6];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int off = ws.nb[row0 + 16 * r + (lane_o >> 2)] * 384 + 32 * (lane_o & 3);
                pj[r][0] = bufld4(r_p, off); pj[r][1] = bufld4(r_p, off + 16);
                pj[r][2] = bufld4(r_p, off + 128); pj[r][3] = bufld4(r_p, off + 144);
                pj[r][4] = bufld4(r_p, off + 256); pj[r][5] = bufld4(r_p, off + 272);
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int e = 16 * r + (lane_o >> 2);
                const float rx = ws.geo[0][row0 + e], ry = ws.geo[1][row0 + e], rz = ws.geo[2][row0 + e];
                const f32x4 a = pj[r][0] * rx + pj[r][2] * ry + pj[r][4] * rz;
                const f32x4 b = pj[r][1] * rx + pj[r][3] * ry + pj[r][5] * rz;
                f16x8 fh, fl;
                split8(a, b, fh, fl);
                float* dst = ws.stage + e * ST32 + 8 * (lane_o & 3);
                *reinterpret_cast<f16x8*>(dst) = fh;
                *reinterpret_cast<f16x8*>(dst + 4) = fl;
            }
        }
        __builtin_amdgcn_wave_barrier();
        f16x8 prh[2], prl[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const float* src = ws.stage + n * ST32 + 8 * (2 * ks + hl);
            prh[ks] = ld8h(src); prl[ks] = ld8h(src + 4);
        }
        __builtin_amdgcn_wave_barrier();
        // first layer of two 32-feature blocks rb0, rb0 + 1: accumulators start as A_j (staging), centre MFMA, W1P MFMAs, ELU, split
        f16x8 h1h[2][2], h1l[2][2];
        auto layer1_pair = [&](auto RB0) {
            constexpr int rb0 = decltype(RB0)::value;
            f32x16 acc[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int rb = rb0 + k;
                // A_j chunk of this block: producer lanes -> staging -> accumulator registers
#pragma unroll
                for (int i = 0; i < 4; ++i) st4(ws.stage + (8 * i + (lane >> 3)) * ST32 + 4 * (lane & 7), aj[k][i]);
                __builtin_amdgcn_wave_barrier();
                if (rb0 == 0) {      // the chunks of blocks 2, 3 fly during the key networks
#pragma unroll
                    for (int i = 0; i < 4; ++i) aj[k][i] = bufld4(r_nb, aj_off[i] + 128 * (rb + 2));
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = ld4(ws.stage + n * ST32 + 4 * (2 * q + hl));
                    acc[k][4 * q] = v[0]; acc[k][4 * q + 1] = v[1]; acc[k][4 * q + 2] = v[2]; acc[k][4 * q + 3] = v[3];
                }
                __builtin_amdgcn_wave_barrier();
            }
            // centre operands: (G0 G1 G2 U) hi, then lo (hl = 0) or the distance weights (hl = 1)
#pragma unroll
            for (int cc = 0; cc < CPT; ++cc) {
                u32x4 acen[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int rb = rb0 + k;
                    unsigned h01, l01, h2u, l2u;
                    split2(cg[cc][rb][0], cg[cc][rb][1], h01, l01);
                    split2(cg[cc][rb][2], cg[cc][rb][3], h2u, l2u);
                    if (hl == 1) {
                        const float* wd = smw + EL32_WD + 2 * (32 * rb + n);
                        l01 = __builtin_bit_cast(unsigned, wd[0]); l2u = __builtin_bit_cast(unsigned, wd[1]);
                    }
                    acen[k] = u32x4{h01, h2u, l01, l2u};
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) acc[k] = MFMA32(__builtin_bit_cast(f16x8, acen[k]), __builtin_bit_cast(f16x8, bcen[cc]), acc[k]);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                f16x8 wh[2], wl[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) { wh[k] = frag32(smw + EL32_W1P, ((rb0 + k) * 2 + ks) * 2, lane); wl[k] = frag32(smw + EL32_W1P, ((rb0 + k) * 2 + ks) * 2 + 1, lane); }
#pragma unroll
                for (int k = 0; k < 2; ++k) acc[k] = MFMA32(wh[k], prh[ks], acc[k]);
#pragma unroll
                for (int k = 0; k < 2; ++k) acc[k] = MFMA32(wh[k], prl[ks], acc[k]);
#pragma unroll
                for (int k = 0; k < 2; ++k) acc[k] = MFMA32(wl[k], prh[ks], acc[k]);
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                sat_probe(sat, acc[k][0]);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const f32x4 u0 = elu4s(f32x4{acc[k][8 * ks], acc[k][8 * ks + 1], acc[k][8 * ks + 2], acc[k][8 * ks + 3]});
                    const f32x4 u1 = elu4s(f32x4{acc[k][8 * ks + 4], acc[k][8 * ks + 5], acc[k][8 * ks + 6], acc[k][8 * ks + 7]});
                    split8(u0, u1, h1h[k][ks], h1l[k][ks]);
                }
            }
        };
        layer1_pair(std::integral_constant<int, 0>{});
        P32_MARK(2);
        // ------------------------------------------------------------------ key networks (eqkm | epkm layers 2, 3) -> logits
        float la[2], lb[2];
        {
            f32x16 a2[2];
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = ld4(smw + EL32_B2 + 32 * k + 4 * (2 * q + hl));
                    a2[k][4 * q] = v[0]; a2[k][4 * q + 1] = v[1]; a2[k][4 * q + 2] = v[2]; a2[k][4 * q + 3] = v[3];
                }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                f16x8 wh[2], wl[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) { wh[k] = frag32(smw + (k ? EL32_W2EP : EL32_W2EQ), ks * 2, lane); wl[k] = frag32(smw + (k ? EL32_W2EP : EL32_W2EQ), ks * 2 + 1, lane); }
#pragma unroll
                for (int k = 0; k < 2; ++k) a2[k] = MFMA32(wh[k], h1h[k][ks], a2[k]);
#pragma unroll
                for (int k = 0; k < 2; ++k) a2[k] = MFMA32(wh[k], h1l[k][ks], a2[k]);
#pragma unroll
                for (int k = 0; k < 2; ++k) a2[k] = MFMA32(wl[k], h1h[k][ks], a2[k]);
            }
            f16x8 h2h[2][2], h2l[2][2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                sat_probe(sat, a2[k][0]);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const f32x4 u0 = elu4s(f32x4{a2[k][8 * ks], a2[k][8 * ks + 1], a2[k][8 * ks + 2], a2[k][8 * ks + 3]});
                    const f32x4 u1 = elu4s(f32x4{a2[k][8 * ks + 4], a2[k][8 * ks + 5], a2[k][8 * ks + 6], a2[k][8 * ks + 7]});
                    split8(u0, u1, h2h[k][ks], h2l[k][ks]);
                }
            }
            // keys: rows 4 part + kappa, K = [eq h2 | ep h2]: two partial accumulators (the eq and the ep half), summed
            f32x16 ka[2];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = ld4(smw + EL32_BK + 4 * (2 * q + hl));
                ka[0][4 * q] = v[0]; ka[0][4 * q + 1] = v[1]; ka[0][4 * q + 2] = v[2]; ka[0][4 * q + 3] = v[3];
                ka[1][4 * q] = 0.f; ka[1][4 * q + 1] = 0.f; ka[1][4 * q + 2] = 0.f; ka[1][4 * q + 3] = 0.f;
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                f16x8 wh[2], wl[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) { wh[k] = frag32(smw + EL32_W3K, (2 * k + ks) * 2, lane); wl[k] = frag32(smw + EL32_W3K, (2 * k + ks) * 2 + 1, lane); }
#pragma unroll
                for (int k = 0; k < 2; ++k) ka[k] = MFMA32(wh[k], h2h[k][ks], ka[k]);
#pragma unroll
                for (int k = 0; k < 2; ++k) ka[k] = MFMA32(wh[k], h2l[k][ks], ka[k]);
#pragma unroll
                for (int k = 0; k < 2; ++k) ka[k] = MFMA32(wl[k], h2h[k][ks], ka[k]);
            }
            // lane (n, hl): registers 0..2 = key of part hl, 4..6 = key of part 2 + hl (kappa = 0..2) of edge n
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                la[h] = (QA[3 * h] * (ka[0][0] + ka[1][0]) + QA[3 * h + 1] * (ka[0][1] + ka[1][1]) + QA[3 * h + 2] * (ka[0][2] + ka[1][2])) * inv_sdk;
                lb[h] = (QV[3 * h] * (ka[0][4] + ka[1][4]) + QV[3 * h + 1] * (ka[0][5] + ka[1][5]) + QV[3 * h + 2] * (ka[0][6] + ka[1][6])) * inv_sdk;
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) { DBG32((t * 8 + h) * 64 + lane, la[h]); DBG32((t * 8 + 2 + h) * 64 + lane, lb[h]); }
        // ------------------------------------------------------------------ softmax of the tile (:139-140), unnormalised weights
        // scalar: part 0 (hl = 0, la); vector: parts 1 (hl = 1, la), 2 (hl = 0, lb), 3 (hl = 1, lb) share one softmax over 3 nn slots
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float mA = cmax(la[h]), mB = cmax(lb[h]);
            float mv = xhalf<true>(hl ? fmaxf(mA, mB) : mB);
            float ms = mA;                                     // (meaningful on the hl = 0 lanes)
            if (NN == 64) {
                ms = lane_bcast(mA, 0);                        // one centre per wave: wave-uniform
                if (t > 0) {
                    const float nms = fmaxf(run_ms[h], ms), nmv = fmaxf(run_mv[h], mv);
                    const float as = __builtin_amdgcn_exp2f((run_ms[h] - nms) * L2E), av = __builtin_amdgcn_exp2f((run_mv[h] - nmv) * L2E);
                    zq[0][h] *= as; S_s[h] *= as;
                    zp1[0][h][0] *= av; zp1[0][h][1] *= av; zp1[0][h][2] *= av; z3[h] *= av; S_v[h] *= av; W2[h] *= av;
                    ms = nms; mv = nmv;
                }
                run_ms[h] = ms; run_mv[h] = mv;
            }
            const float ea = __builtin_amdgcn_exp2f((la[h] - (hl ? mv : ms)) * L2E);
            const float eb = __builtin_amdgcn_exp2f((lb[h] - mv) * L2E);
            const float sA = csum(ea), sB = csum(eb);
            const float sv = xhalf<false>(hl ? sA + sB : sB);
            if (NN == 64) { S_s[h] += sA; S_v[h] += sv; W2[h] += sB; }
            else { S_s[h] = sA; S_v[h] = sv; W2[h] = sB; }
            DBG32((t * 8 + 4 + h) * 64 + lane, ea); DBG32((t * 8 + 6 + h) * 64 + lane, eb);
            if (hl == 0) {
                ws.wt[h][n] = ea;
            } else {
                ws.wt[2 + 3 * h][n] = ea * gx; ws.wt[3 + 3 * h][n] = ea * gy; ws.wt[4 + 3 * h][n] = ea * gz;
                ws.wt[8 + h][n] = eb;
            }
        }
        // statistics of the centres that end with this tile: written by the first lane of the centre (hl = 0 holds s_scalar and the part-2 sum)
        if ((NN != 64 || t == T32 - 1) && hl == 0 && (n & (NN == 16 ? 15 : 31)) == 0) {
            const int slot = NN == 64 ? 0 : NN == 32 ? t : (n >> 4);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float iv = __builtin_amdgcn_rcpf(S_v[h]);
                ws.stat[slot][h][0] = __builtin_amdgcn_rcpf(S_s[h]); ws.stat[slot][h][1] = iv; ws.stat[slot][h][2] = W2[h] * iv;
            }
        }
        __builtin_amdgcn_wave_barrier();
        P32_MARK(4);
        // ------------------------------------------------------------------ p_j sums, first half of the tile's edges issued now
        f32x4 pv[8];
        {
            const u32x4 nb4a = *reinterpret_cast<const u32x4*>(&ws.nb[row0 + 16 * es]), nb4b = *reinterpret_cast<const u32x4*>(&ws.nb[row0 + 16 * es + 4]);
#pragma unroll
            for (int i = 0; i < 4; ++i) { pv[i] = bufld4(r_p, (int)nb4a[i] * 384 + 16 * quad); pv[4 + i] = bufld4(r_p, (int)nb4b[i] * 384 + 16 * quad); }
        }
        // ------------------------------------------------------------------ value network (evm), first layer blocks 2, 3
        f16x8 vh[2][2], vl[2][2];
        {
            // (h1h / h1l are reused for the value half)
            layer1_pair(std::integral_constant<int, 2>{});
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) { vh[k][ks] = h1h[k][ks]; vl[k][ks] = h1l[k][ks]; }
        }

        f32x16 a2[2];
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = ld4(smw + EL32_B2 + 64 + 32 * k + 4 * (2 * q + hl));
                a2[k][4 * q] = v[0]; a2[k][4 * q + 1] = v[1]; a2[k][4 * q + 2] = v[2]; a2[k][4 * q + 3] = v[3];
            }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {       // k-steps of the 64 inputs: block kk >> 1, step kk & 1
            f16x8 wh[2], wl[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) { wh[k] = frag32(smw + EL32_W2EV, (k * 4 + kk) * 2, lane); wl[k] = frag32(smw + EL32_W2EV, (k * 4 + kk) * 2 + 1, lane); }
#pragma unroll
            for (int k = 0; k < 2; ++k) a2[k] = MFMA32(wh[k], vh[kk >> 1][kk & 1], a2[k]);
#pragma unroll
            for (int k = 0; k < 2; ++k) a2[k] = MFMA32(wh[k], vl[kk >> 1][kk & 1], a2[k]);
#pragma unroll
            for (int k = 0; k < 2; ++k) a2[k] = MFMA32(wl[k], vh[kk >> 1][kk & 1], a2[k]);
        }
        P32_MARK(6);
        // first half of the p_j sums (their loads had the first two value layers to land)
        {
            const float* w3a = &ws.wt[8][16 * es], *w3b = &ws.wt[9][16 * es];
            const f32x4 wa0 = ld4(w3a), wa1 = ld4(w3a + 4), wb0 = ld4(w3b), wb1 = ld4(w3b + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) { z3[0] += wa0[i] * pv[i]; z3[1] += wb0[i] * pv[i]; z3[0] += wa1[i] * pv[4 + i]; z3[1] += wb1[i] * pv[4 + i]; }
            const u32x4 nb4a = *reinterpret_cast<const u32x4*>(&ws.nb[row0 + 16 * es + 8]), nb4b = *reinterpret_cast<const u32x4*>(&ws.nb[row0 + 16 * es + 12]);
#pragma unroll
            for (int i = 0; i < 4; ++i) { pv[i] = bufld4(r_p, (int)nb4a[i] * 384 + 16 * quad); pv[4 + i] = bufld4(r_p, (int)nb4b[i] * 384 + 16 * quad); }
        }
        f16x8 g2h[2][2], g2l[2][2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            sat_probe(sat, a2[k][0]);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const f32x4 u0 = elu4s(f32x4{a2[k][8 * ks], a2[k][8 * ks + 1], a2[k][8 * ks + 2], a2[k][8 * ks + 3]});
                const f32x4 u1 = elu4s(f32x4{a2[k][8 * ks + 4], a2[k][8 * ks + 5], a2[k][8 * ks + 6], a2[k][8 * ks + 7]});
                split8(u0, u1, g2h[k][ks], g2l[k][ks]);
            }
        }
        // values V[edge][feature]: edges are the MFMA rows (A operand = h2), weights the B operand; column block 0 = q values, 1 = p values
        f32x16 vv[2];
        {
            const float bp = smw[EL32_B3V + 32 + n];
#pragma unroll
            for (int i = 0; i < 16; ++i) { vv[0][i] = 0.f; vv[1][i] = bp; }      // (the q block's bias is added after the normalisation: its weights sum to 1)
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            f16x8 bh[2], bl[2];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) { bh[cb] = frag32(smw + EL32_W3V, (cb * 4 + kk) * 2, lane); bl[cb] = frag32(smw + EL32_W3V, (cb * 4 + kk) * 2 + 1, lane); }
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) vv[cb] = MFMA32(g2h[kk >> 1][kk & 1], bh[cb], vv[cb]);
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) vv[cb] = MFMA32(g2l[kk >> 1][kk & 1], bh[cb], vv[cb]);
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) vv[cb] = MFMA32(g2h[kk >> 1][kk & 1], bl[cb], vv[cb]);
        }
        // second half of the p_j sums
        {
            const float* w3a = &ws.wt[8][16 * es + 8], *w3b = &ws.wt[9][16 * es + 8];
            const f32x4 wa0 = ld4(w3a), wa1 = ld4(w3a + 4), wb0 = ld4(w3b), wb1 = ld4(w3b + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) { z3[0] += wa0[i] * pv[i]; z3[1] += wb0[i] * pv[i]; z3[0] += wa1[i] * pv[4 + i]; z3[1] += wb1[i] * pv[4 + i]; }
        }
        // attention-weighted sums over this lane's sixteen edges (registers 4 q + r <-> edge 8 q + 4 hl + r)  (:143-144, first block of Vp :132)
        // one weight row at a time: its four LDS reads (the lane's sixteen edges; broadcast within a half) are issued one row ahead
        {
            f32x4 wr[2][4];
            auto load_row = [&](int row, f32x4* dst) {
#pragma unroll
                for (int q = 0; q < 4; ++q) dst[q] = ld4(&ws.wt[row][8 * q + 4 * hl]);
            };
            load_row(0, wr[0]);
#pragma unroll
            for (int row = 0; row < 8; ++row) {       // rows 0, 1: scalar weights of head 0, 1 (q values); 2 + 3 h + c: part-1 weights x r_c (p values)
                if (row < 7) load_row(row + 1, wr[(row + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                const f32x4* w = wr[row & 1];
                const int cb = row < 2 ? 0 : 1;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int cs = NN == 16 ? (q >> 1) : 0;
                    float& z = row < 2 ? zq[cs][row] : zp1[cs][(row - 2) / 3][(row - 2) % 3];
#pragma unroll
                    for (int r = 0; r < 4; ++r) z = __builtin_fmaf(w[q][r], vv[cb][4 * q + r], z);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        P32_MARK(9);
        if (NN == 64 && t == 0) continue;      // the centre continues in the second tile
        // ------------------------------------------------------------------ centre(s) complete: combine, normalise, stage the Z rows
        {   // p_j sums -> staging area [slot][h][96] (nn = 16: the two edge halves ARE the two centres; otherwise fold them)
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (NN != 16) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) z3[h][j] += __shfl_down(z3[h][j], 24);
                }
                if (NN == 16 ? lane < 48 : esub == 0) st4(ws.stage + ((NN == 16 ? es : 0) * 2 + h) * 96 + 4 * quad, z3[h]);
                z3[h] = f32x4{0, 0, 0, 0};
            }
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int cs = 0; cs < NCS; ++cs) {
            const int slot = NN == 64 ? sub : NN == 32 ? t : cs;       // staged row of this centre (FIN)
            const int sslot = NN == 64 ? 0 : NN == 32 ? t : cs;        // where its statistics are
            float* zb = zrow[slot];
            // totals over the two lane halves; this lane then finishes head h = hl
            float tq[2], tp[2][3];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                tq[h] = xhalf<false>(zq[cs][h]);
#pragma unroll
                for (int c = 0; c < 3; ++c) tp[h][c] = xhalf<false>(zp1[cs][h][c]);
                zq[cs][h] = 0.f; zp1[cs][h][0] = zp1[cs][h][1] = zp1[cs][h][2] = 0.f;
            }
            const f32x4 st = ld4(&ws.stat[sslot][hl][0]);
            DBG32(1024 + 4 * lane, st[0]); DBG32(1024 + 4 * lane + 1, st[1]); DBG32(1024 + 4 * lane + 2, st[2]); DBG32(1024 + 4 * lane + 3, (float)cs);
            DBG32(1280 + cs * 512 + lane, tq[0]); DBG32(1280 + cs * 512 + 64 + lane, tq[1]);
#pragma unroll
            for (int c = 0; c < 3; ++c) { DBG32(1280 + cs * 512 + 128 + c * 64 + lane, tp[0][c]); DBG32(1280 + cs * 512 + 320 + c * 64 + lane, p_own[cs][c]); }
            const float q_own = hl ? tq[1] : tq[0];
            zb[hl * 32 + n] = q_own * st[0] + smw[EL32_B3V + n];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float p_sum = hl ? tp[1][c] : tp[0][c];
                zb[64 + c * 64 + hl * 32 + n] = (p_sum + ws.stage[((NN == 16 ? cs : 0) * 2 + hl) * 96 + c * 32 + n]) * st[1] + st[2] * p_own[cs][c];
            }
        }
        __builtin_amdgcn_wave_barrier();
#ifdef PESTO_DEBUG32
        if (dbg_) for (int cs = 0; cs < NCS; ++cs) for (int k = lane; k < 256; k += 64) g_dbg32[2304 + cs * 256 + k] = zrow[NN == 64 ? sub : NN == 32 ? t : cs][k];
#endif
        P32_MARK(10);
    }
}

// WPB = waves per workgroup: 4 (exact fp32 path: two workgroups per CU, 2 waves/SIMD, explicit cross-tile prefetch PF)
// or 12 / 8 (f16-split path: one workgroup per CU, 3 / 2 waves per SIMD sharing one LDS copy of the layer constants).
// FIN (finish in the edge kernel): the attention sums Z of a centre never leave the CU. Every wave leaves the complete Z rows of
// its (at most two) centres in its LDS scratch; behind a workgroup barrier four waves per 16 centres apply the output MLPs on
// the matrix cores - role 0: q += qpm(Zq), roles 1..3: p[c] += ppm(Zp[c]) (model_operations.py:147-152), sink reset (:239-240) -
// with the weight fragments streamed from L2, and write the NEW state into the other half of a ping-pong pair (neighbours'
// p_j of the old state are still being gathered by other workgroups). The node kernel then only prepares records.
// s_setprio takes an immediate: one scalar branch per age class (the class is wave-uniform, in an SGPR)
template <int P0, int P1, int P2>
__device__ __forceinline__ void prio_by_age(int cls) {
    if constexpr (P0 == P1 && P1 == P2) { __builtin_amdgcn_s_setprio(P0); }
    else {
        if (cls == 0) __builtin_amdgcn_s_setprio(P0);
        else if (cls == 1) __builtin_amdgcn_s_setprio(P1);
        else __builtin_amdgcn_s_setprio(P2);
    }
}
template <int NN, int WPB, bool PF, bool F16, bool HY = false, int TI = 4, bool FIN = false, int NE = WPB, bool M32 = false>
__global__ __launch_bounds__(WPB * 64, WPB == 4 ? 2 : WPB / 4) void k_edge(const float* __restrict__ W, LayerW lw, int N1, int n_work,
                                                 const int* __restrict__ ids_s, const float4* __restrict__ geo,
                                                 const float* __restrict__ rec_nb, const float* rec_cen,
                                                 const float* __restrict__ p_state, float* __restrict__ Z, int* __restrict__ flags,
                                                 const float* __restrict__ q_state, float* __restrict__ q_out, float* __restrict__ p_out,
                                                 PrepW lwp, float* __restrict__ rec_nb_out, float* rec_cen_out) {
    // rec_cen_out may be rec_cen itself: a centre's record is read only by the wave that processes the centre, before the finish phase
    // of the same workgroup iteration rewrites it (no __restrict__ on the pair); rec_nb is gathered by every workgroup: separate buffers
    // TI = 16-edge tiles per wave work item: 4 (64 edge rows) for full launches; small launches (one structure) use finer
    // items - 1 tile for nn = 8 / 16, 2 for nn = 32 - so that the launch is spread over more waves and CUs (latency)
    constexpr int A = 16 * TI / NN;            // whole centres per work item
    constexpr int TPC = NN >= 16 ? NN / 16 : 1;   // tiles per centre
    static_assert(TI >= TPC && TI % TPC == 0 && A >= 1 && (!PF || TI == 4), "a work item holds whole centres");
    static_assert(!HY || (F16 && !PF), "the hybrid first layer exists on the lean f16-split path only");
    static_assert(HY == F16, "the f16-split tables are in the log2 domain of the hybrid path (no full-record f16 twin any more)");
    static_assert(!FIN || (HY && A <= 2 && WPB >= 8), "finish phase: at most two staged centres per wave, four waves per 16-centre tile");
    // FIN: work items a wave processes between two finish phases - as many as its two staging rows hold centres (nn = 64: two
    // one-centre items), which halves the number of workgroup rendezvous
    constexpr int SUBS = FIN ? 2 / A : 1;
    constexpr bool NODEW = FIN && NE < WPB;      // node-wave mode: waves NE.. finish / prepare only
    // node-wave mode: four node waves; eight item waves (one 16-centre tile per iteration) or - sixteen-wave workgroups, one-tile items,
    // 128 registers: FOUR waves per SIMD, three of them on work items - twelve (24 centres per iteration: one full tile + one half tile)
    static_assert(NE == WPB || (FIN && WPB - NE == 4 && (NE * A * SUBS == 16 || (NE == 12 && TI == 1 && A * SUBS == 2))) || (FIN && WPB - NE == 8 && NE * A * SUBS == 16),
                  "node waves: four of them (16 or 24 centres per iteration), or two teams of four that take the iterations in turn");
    constexpr int NTEAM = (FIN && NE < WPB) ? (WPB - NE) / 4 : 1;
    constexpr int NWT = NODEW ? (NE * A * SUBS + 15) / 16 : 1;      // 16-centre tiles of an iteration (node-wave mode)
    // TAILR (round 5, measured and NOT shipped; -DPESTO_TAILR builds it): the LAST iteration's tile of a node-wave workgroup finished by its
    // eight ITEM waves behind a rendezvous - the finish / prepare phase of the eight-wave rendezvous kernels (fragments requested in front
    // of the barrier, four finishing + four [U|A] waves) instead of the four node waves' chain that every node-wave launch ends with while
    // its item waves have already left. Same bits (49 parity tests incl. the bitwise mode equalities) - and no gain: four runs each on
    // one box, nn = 8 / 16 / 32 55.8 / 85.2 / 143.9 -> 55.8 / 84.4 / 142.7 us per launch, the step 1,829 -> 1,820 structures/s
    // (profiles/r05_node_loop_ab.txt): with the SIMDs to themselves the node waves' last chain is short, and the rendezvous waits for them to end.
#ifdef PESTO_TAILR
    constexpr bool TAILR = NODEW && NE == 8 && WPB - NE == 4 && NE * A * SUBS == 16;
#else
    constexpr bool TAILR = false;
#endif
    constexpr int WROWS = (NODEW && (NE > 8 || WPB - NE == 8)) ? 16 * TI : 64;          // rows of the per-wave scratch (sixteen-wave workgroups: LDS)
    static_assert(!M32 || (HY && NE == WPB && WPB == 8 && NN >= 16 && TI % 2 == 0), "M32: eight-wave workgroups, whole 32-edge tiles");
    __shared__ EdgeSmem<WPB, HY, FIN, NE, M32, WROWS> sm;
    int tr_n = 0;           // (developer builds: timeline of one wave)
    (void)tr_n;
    TRACE32(NN == 8 ? 60 : NN == 16 ? 61 : NN == 32 ? 62 : 63);
    if (threadIdx.x < 8) sm.xflag[threadIdx.x] = 0;
    // The edge rows (neighbour id, geometry) of the wave's FIRST work item are requested before the layer constants are staged, so that
    // their round trip runs under that copy instead of behind the workgroup barrier - a small launch (one structure) is one or two
    // items per wave deep and pays every such latency in full (one-structure forward: see DESIGN 4.1f).
    int first_nb = 0;
    float4 first_geo = float4{0.f, 0.f, 0.f, 0.f};
    bool first_valid = false, first_item = false;
    if constexpr (!M32) {
        const int xcd0 = blockIdx.x & 7, jb0 = blockIdx.x >> 3, nbx0 = gridDim.x >> 3, chunk0 = (n_work + 7) >> 3;
        const int w_end0 = min(n_work, (xcd0 + 1) * chunk0), it0 = xcd0 * chunk0;
        const bool tail0 = w_end0 - it0 < nbx0 * NE * SUBS;
        const int lane0 = threadIdx.x & 63, wave0 = threadIdx.x >> 6;
        const int work0 = it0 + jb0 * NE * (tail0 ? 1 : SUBS) + wave0;
        first_item = work0 < w_end0 && (!NODEW || wave0 < NE);
        if (first_item) {
            const int i0 = work0 * A + lane0 / NN;
            const size_t src0 = (size_t)min(i0, N1 - 1) * KMAX + lane0 % NN;
            first_valid = i0 < N1 && lane0 < 16 * TI;
            first_nb = ids_s[src0];
            first_geo = geo[src0];
        }
    }
    {   // layer constants -> LDS (once per workgroup; workgroups are persistent over work items)
        const f32x4* src = reinterpret_cast<const f32x4*>(W + (M32 ? lw.e_lds32 : F16 ? lw.e_lds16 : lw.e_lds));
        copy_to_lds<(M32 ? EDGE_LDS_FLOATS_32 : HY ? EDGE_LDS_FLOATS_HY : EDGE_LDS_FLOATS) / 4, WPB * 64>(reinterpret_cast<f32x4*>(sm.w), src, (int)threadIdx.x);
    }
    if constexpr (HY && !M32) {
        if (threadIdx.x < 64) {
            const float b = W[lw.e_lds16 + EL_B3V + threadIdx.x];
            st4(&sm.b3v4[4 * threadIdx.x], f32x4{b, b, b, b});
        }
    }
    if constexpr (!M32) {
        if (first_item) {     // rows of the first item -> the wave's scratch (no register survives into the work loop)
            auto& ws0 = sm.ws[threadIdx.x >> 6];
            const int l0 = threadIdx.x & 63;
            if (WROWS == 64 || l0 < WROWS) {
                ws0.nb[l0] = first_valid ? first_nb : 0;
                ws0.geo[0][l0] = first_valid ? first_geo.x : 0.f; ws0.geo[1][l0] = first_valid ? first_geo.y : 0.f;
                ws0.geo[2][l0] = first_valid ? first_geo.z : 0.f; ws0.geo[3][l0] = first_valid ? first_geo.w : 0.f; ws0.geo[4][l0] = 1.0f;
            }
        }
    }
    __syncthreads();
    TRACE32(50);
    const float* w2f = sm.w + EL_W2F;
    const float* w3k = sm.w + EL_W3K;
    const float* w3v = sm.w + EL_W3V;
    const float inv_sdk = 1.0f / sqrtf((float)NK);   // logits / sdk (model_operations.py:139-140) as a multiply

    // XCD-aware work mapping: workgroup b runs on XCD b % 8 (observed dispatch order; speed only). Each XCD owns
    // one contiguous eighth of the work items, consecutive workgroups of an XCD take consecutive items, so the
    // neighbour records a CU gathers are mostly ones its own XCD's L2 already holds.
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3, nbx = gridDim.x >> 3;
    const int chunk = (n_work + 7) >> 3;
    const int w_end = min(n_work, (xcd + 1) * chunk);
    PHASE_DECL();
    // range guard of the f16-split path (sat_probe), flushed per centre (sat_flush_at). sat_b: the second centre of a two-centre item
    // whose centres are different tiles (nn = 16 / 32); for nn = 8 the two centres of a tile are different lanes
    float sat = 0.0f, sat_b = 0.0f;
    constexpr bool SAT2 = F16 && FIN && !M32 && A == 2 && NN >= 16;
    int fin_iter = 0;       // finish phases done (FIN)
    const bool only_fl = (!F16 && !FIN) ? only_flagged_of(flags) : false;      // (SatCtx::only_flagged: read once, uniform)
    (void)only_fl;
    // the trip count is the same for every wave of a workgroup (FIN: workgroup barriers inside); a wave without a work item idles
    // An iteration hands every wave SUBS items. Full iterations: consecutive blocks of WPB items per wave-slot (neighbouring centres
    // share gathered lines in the L1). The LAST iteration of an XCD's share (fewer items left than slots - in a small launch the only
    // one): the second items start behind the first items of ALL workgroups, so that the remainder is spread over the workgroups
    // instead of giving a few of them two items per wave and the rest none. Same trip count for every workgroup of the XCD.
    if constexpr (NODEW) {
      if (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) >= NE) {
        // ======== the node waves' own loop (same iteration arithmetic as the item waves' loop below; separate since round 5, so that the
        // two kinds of waves are two register-allocation problems: the item loop's live values no longer sit under the node chain)
        // ---- node-wave mode: the same finish + prepare arithmetic, but by four waves that do nothing else, fed through LDS queues.
        // An edge wave stages the Z rows of its two centres of this iteration in generation (iteration & 1) of its staging rows,
        // counts itself in XF_READY[generation] and goes straight on to its next items: no rendezvous, no weight fragments, no
        // record stores (whose acknowledgements the next gathers of the same wave would have to wait for) on the waves that
        // carry the edge work. Node wave r (role r: 0 = q, c + 1 = p[c]) waits for the eight edge waves, copies its part of the 16
        // rows to registers, counts itself in XF_CONSUMED (two iterations later the edge waves overwrite the generation), runs its
        // finish chain, posts its slice of the new tile state (exchange buffer, also two generations), then G[c] (roles 1..3), and
        // - once all four slices are posted - Q (role 0) and the [U|A] blocks 4r..4r+3.
        constexpr int CPW = A * SUBS;
        const int lane = threadIdx.x & 63;
        const int wave_u = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        for (int it_start = xcd * chunk; it_start < w_end; it_start += nbx * NE * SUBS) {
            if (TAILR && it_start + nbx * NE * SUBS >= w_end) break;      // the last iteration's tile: the item waves' (behind their loop)
            const bool tail = w_end - it_start < nbx * NE * SUBS;
            const int sstride = tail ? nbx * NE : NE;
            const int base = it_start + jb * NE * (tail ? 1 : SUBS);
            const int gen = fin_iter & 1;
            (void)sstride;
            // weight fragments of the node waves: plain global loads, or (-DPESTO_NODEW_WAUX=<aux>, developer) buffer loads with cache-policy bits
            // (gfx950: 1 = sc0, 2 = nt, 16 = sc1) - the 166 KB stream per 16 centres goes through the 32 KB L1 the gathers live in
#ifdef PESTO_NODEW_WAUX
            const __amdgpu_buffer_rsrc_t rsW = make_rsrc(W);
#define LDW(ptr) __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rsW, (int)(((ptr) - W) * 4), 0, PESTO_NODEW_WAUX))
#else
#define LDW(ptr) ld8h(ptr)
#endif
            // (two teams: team t takes the iterations of generation t - a tile's chain may then last two iterations)
            const int role = (wave_u - NE) & 3;
            const bool prep = rec_cen_out != nullptr;
            const int tq = NTEAM == 2 ? gen : 0;          // counters / exchange buffer of this team
#pragma unroll 1
          for (int ntile = 0; ntile < ((NTEAM == 2 && ((wave_u - NE) >> 2) != gen) ? 0 : NWT); ++ntile) {
            // (the lane index is re-derived per tile from an opaque copy - as invariants of the node waves' loop the weight fragments of the
            // whole phase would be loaded once in front of it and live across it)
            int lane_n = lane;
            asm volatile("" : "+v"(lane_n));
            const int fe = lane_n & 15, fg = lane_n >> 4;
            // centre of MFMA column fe: edge wave fe / CPW (+ 16 / CPW per tile), its staged row fe % CPW (work-item arithmetic of the loop above)
            const int cw_raw = ntile * (16 / CPW) + fe / CPW, cr = fe % CPW;
            const int cw = NWT > 1 ? min(cw_raw, NE - 1) : cw_raw;
            const int cwork = base + (SUBS > 1 ? cr * sstride : 0) + cw;
            const int ci_raw = cwork * A + (SUBS > 1 ? 0 : cr);
            const bool valid = cwork < w_end && ci_raw < N1 && cw_raw < NE;
            const int ci = valid ? ci_raw : 0;
            const float* zr = sm.zrows[cw][gen][cr] + (role == 0 ? 0 : 64 + (role - 1) * 64);
            const float st_limit = state_limit_of(flags);      // conditioning trigger
            const float* fb = W + lw.h_q0 + lane_n * 4;       // fragments q0 | q1 | q2 | pp contiguous in the image, 256 floats each
            const int ntile_seq = fin_iter * NWT + ntile;   // tiles this workgroup's node waves have taken before this one
            // exchange buffer, two generations (a wave posts its next tile while a slower wave of its team still reads this one)
            float* xs = sm.xch + (NTEAM == 2 ? 2 * tq + ((fin_iter >> 1) & 1) : (ntile_seq & 1)) * 2048;    // [q0 q1 p00 p01 p10 p11 p20 p21][fg 4][column 16][4]
            float* cen = rec_cen_out + (size_t)ABL_ST(ci) * REC_CEN;
            f16x8 zh[2], zl[2];
            auto rows = [&]() {                 // wait for the tile's edge waves, then this role's part of the 16 rows as hi/lo B operands
                if (NWT > 1 && ntile == 1) lds_wait_ge<16>(&sm.xflag[XF_READY2 + gen], (NE - 8) * ((fin_iter >> 1) + 1));
                else lds_wait_ge<16>(&sm.xflag[XF_READY + gen], (NE < 8 ? NE : 8) * ((fin_iter >> 1) + 1));
#pragma unroll
                for (int kgp = 0; kgp < 2; ++kgp) {
                    f32x4 a0 = ld4(zr + 32 * kgp + 4 * fg), a1 = ld4(zr + 32 * kgp + 16 + 4 * fg);
                    if (!valid) { a0 = f32x4{0, 0, 0, 0}; a1 = a0; }     // unused columns: no stale LDS bits into the range guard
                    split8(a0, a1, zh[kgp], zl[kgp]);
                }
                lds_signal(&sm.xflag[XF_CONSUMED + tq], lane_n == 0);
#ifdef PESTO_NODEW_PRIO      // (developer: priority of a node wave while it computes; it polls at priority 0)
                __builtin_amdgcn_s_setprio(PESTO_NODEW_PRIO);
#endif
            };
            auto post = [&](const f32x4* v) {
                st4(xs + ((2 * role) * 4 + fg) * 64 + fe * 4, v[0]);
                st4(xs + ((2 * role + 1) * 4 + fg) * 64 + fe * 4, v[1]);
                lds_signal(&sm.xflag[XF_POST + tq], lane_n == 0);
            };
#define PESTO_FIN_MFMA(acc, fr, xh_, xl_)                                                   \
    {                                                                                        \
        _Pragma("unroll") for (int m = 0; m < 2; ++m) acc[m] = MFMA16((fr)[2 * m], xh_, acc[m]);     \
        _Pragma("unroll") for (int m = 0; m < 2; ++m) acc[m] = MFMA16((fr)[2 * m], xl_, acc[m]);     \
        _Pragma("unroll") for (int m = 0; m < 2; ++m) acc[m] = MFMA16((fr)[2 * m + 1], xh_, acc[m]); \
    }
#ifdef PESTO_ABL_NONODE      // ablation (results wrong): the node waves only keep the queues moving - what the item waves cost by themselves
            rows();
            sat_probe(sat, __builtin_bit_cast(float, (int)zh[0][0]));
            lds_signal(&sm.xflag[XF_POST + tq], lane_n == 0);
            continue;
#endif
            f32x4 st[2];
            // From here to the end of the tile the two role classes (0: q; 1..3: p[c]) are SEPARATE paths (always-inline lambdas for what they
            // share): the weight fragments one class holds are then not live on the other's path (round 5; the node waves' loop is a register
            // allocation of its own). Role 0 - the longest chain of a tile: qpm's three layers, its [U|A] blocks, nqm's three layers - requests
            // nqm's first-layer fragments in front of the [U|A] products instead of behind them (-DPESTO_NODEW_NOPREFETCH: the old order);
            // all of nqm there, or the G fragments of roles 1..3 ahead of the wait for the staged rows (-DPESTO_NODEW_PREFETCH_G), do not fit
            // 168 registers (62 ... 133 spilled). Same MFMA order per accumulator: same bits.
            auto finish_state = [&]() __attribute__((always_inline)) {
                sat_probe(sat, st[0][0]);          // (unused columns were fed zeros and the sink row's finite state)
                if (valid) mag_flush_at(st[0], st[1], st_limit, flags, ci);
                if (ci == 0) { st[0] = f32x4{0, 0, 0, 0}; st[1] = st[0]; }                               // :239-240 sink
                if (valid) {
                    float* dst = role == 0 ? q_out + (size_t)ci * S : p_out + (size_t)ci * 96 + (role - 1) * 32;
                    st4(dst + 4 * fg, st[0]); st4(dst + 16 + 4 * fg, st[1]);
                }
            };
            // ---- the NEXT layer's records of these 16 centres (k_node16's prepare half, same arithmetic: model_operations.py:103-119)
            const int ob = 4 * role;              // the [U | A] blocks of this wave: 4 role .. 4 role + 3
            f16x8 ua[2][4][2];                    // [kgp][block][hi|lo]
            f32x4 ub[4];
            auto load_ua = [&]() __attribute__((always_inline)) {
                const float* Lua = W + lwp.h_ua + lane_n * 4;          // [m 16][kgp 2][hi|lo][256]
#pragma unroll
                for (int kgp = 0; kgp < 2; ++kgp)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float* fr = Lua + (size_t)(((ob + j) * 2 + kgp) * 2) * 256;
                        ua[kgp][j][0] = LDW(fr); ua[kgp][j][1] = LDW(fr + 256);
                    }
#pragma unroll
                for (int j = 0; j < 4; ++j) ub[j] = ob < 8 ? ld4(W + lwp.n_b1s + 16 * (ob + j) + 4 * fg) : f32x4{0, 0, 0, 0};
            };
            f16x8 xnh[2], xnl[2];
            auto tile_inputs = [&]() __attribute__((always_inline)) {      // once all four slices are posted: [q | ||p||] of the tile as f16 hi/lo B operands (k-group 0 = q, 1 = ||p||)
                lds_wait_ge(&sm.xflag[XF_POST + tq], NTEAM == 2 ? 4 * (fin_iter >> 1) + 4 : 4 * ntile_seq + 4);
                f32x4 q[2], pn[2];
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    f32x4 p3[3];
                    q[m] = ld4(xs + (m * 4 + fg) * 64 + fe * 4);
#pragma unroll
                    for (int c = 0; c < 3; ++c) p3[c] = ld4(xs + ((2 + 2 * c + m) * 4 + fg) * 64 + fe * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) pn[m][r] = norm3_fast(p3[0][r], p3[1][r], p3[2][r]);
                }
                split8(q[0], q[1], xnh[0], xnl[0]);
                split8(pn[0], pn[1], xnh[1], xnl[1]);
            };
            auto ua_products = [&]() __attribute__((always_inline)) {
                f32x4 a[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) a[j] = ub[j];
#pragma unroll
                for (int kgp = 0; kgp < 2; ++kgp) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[j] = MFMA16(ua[kgp][j][0], xnh[kgp], a[j]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[j] = MFMA16(ua[kgp][j][0], xnl[kgp], a[j]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[j] = MFMA16(ua[kgp][j][1], xnh[kgp], a[j]);
                }
                sat_probe(sat, a[0][0]);
                if (valid) {
                    float* nb = rec_nb_out + (size_t)ABL_ST(ci) * REC_A;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (ob < 8) st4_finite(cen + (ob + j) * 64 + 3 * 16 + 4 * fg, a[j]);
                        else st4(nb + (ob + j - 8) * 16 + 4 * fg, a[j]);                   // A_j[16 fb + 4g + r]
                    }
                }
            };
            if (role == 0) {   // qpm: 64 -> 32 -> 32 -> 32 with ELU between            (model_operations.py:147, :151)
                f16x8 w0[2][4], w1[4], w2[4];            // [kgp][(m, hi|lo)], [(m, hi|lo)]
                f32x4 h[2], b1v[2], b2v[2];
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int kgp = 0; kgp < 2; ++kgp) { w0[kgp][2 * m] = LDW(fb + ((m * 2 + kgp) * 2) * 256); w0[kgp][2 * m + 1] = LDW(fb + ((m * 2 + kgp) * 2 + 1) * 256); }
#pragma unroll
                for (int f = 0; f < 4; ++f) { w1[f] = LDW(fb + (8 + f) * 256); w2[f] = LDW(fb + (12 + f) * 256); }
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    h[m] = ld4(W + lw.n_bq0 + 16 * m + 4 * fg); b1v[m] = ld4(W + lw.n_bq1 + 16 * m + 4 * fg); b2v[m] = ld4(W + lw.n_bq2 + 16 * m + 4 * fg);
                    st[m] = ld4(q_state + (size_t)ci * S + 16 * m + 4 * fg);
                }
                rows();
                PESTO_FIN_MFMA(h, w0[0], zh[0], zl[0])
                PESTO_FIN_MFMA(h, w0[1], zh[1], zl[1])
                sat_probe(sat, h[0][0]);
                f16x8 xh, xl;
                split8(elu4(h[0]), elu4(h[1]), xh, xl);
                PESTO_FIN_MFMA(b1v, w1, xh, xl)
                sat_probe(sat, b1v[0][0]);
                split8(elu4(b1v[0]), elu4(b1v[1]), xh, xl);
                PESTO_FIN_MFMA(b2v, w2, xh, xl)
#pragma unroll
                for (int m = 0; m < 2; ++m) st[m] += b2v[m];
                finish_state();
                if (prep) {
                    post(st);
                    load_ua();               // on their way while the other slices are being posted (requested earlier - behind qpm's first layer - they spill 102 registers)
                    // node queries Q = nqm(X_n): 64 -> 32 -> 32 -> 12 (:119): fragments n0 [m 2][kgp 2] | n1 [m 2] | n2 [1], (hi, lo) pairs of 256 floats
                    const float* nq = W + lwp.h_n0 + lane_n * 4;
                    f16x8 n0[2][4], n1[4], n2[2];
                    f32x4 hq[2], tq2[2], qq[1];
                    auto load_nq0 = [&]() __attribute__((always_inline)) {
#pragma unroll
                        for (int m = 0; m < 2; ++m)
#pragma unroll
                            for (int kgp = 0; kgp < 2; ++kgp) { n0[kgp][2 * m] = LDW(nq + ((m * 2 + kgp) * 2) * 256); n0[kgp][2 * m + 1] = LDW(nq + ((m * 2 + kgp) * 2 + 1) * 256); }
#pragma unroll
                        for (int m = 0; m < 2; ++m) hq[m] = ld4(W + lwp.n_bn0 + 16 * m + 4 * fg);
                    };
                    tile_inputs();
#ifndef PESTO_NODEW_NOPREFETCH
                    load_nq0();              // nqm's first layer: requested in front of the [U|A] products (all of nqm here: 168 registers do not hold it)
#endif
                    __builtin_amdgcn_sched_barrier(0);
                    ua_products();
                    __builtin_amdgcn_sched_barrier(0);
#ifdef PESTO_NODEW_NOPREFETCH
                    load_nq0();
#endif
#pragma unroll
                    for (int f = 0; f < 4; ++f) n1[f] = LDW(nq + (8 + f) * 256);
                    n2[0] = LDW(nq + 12 * 256); n2[1] = LDW(nq + 13 * 256);
#pragma unroll
                    for (int m = 0; m < 2; ++m) tq2[m] = ld4(W + lwp.n_bn1 + 16 * m + 4 * fg);
                    qq[0] = ld4(W + lwp.n_bn2 + 4 * fg);
                    PESTO_FIN_MFMA(hq, n0[0], xnh[0], xnl[0])
                    PESTO_FIN_MFMA(hq, n0[1], xnh[1], xnl[1])
                    sat_probe(sat, hq[0][0]);
                    split8(elu4(hq[0]), elu4(hq[1]), xh, xl);
                    PESTO_FIN_MFMA(tq2, n1, xh, xl)
                    sat_probe(sat, tq2[0][0]);
                    split8(elu4(tq2[0]), elu4(tq2[1]), xh, xl);
                    qq[0] = MFMA16(n2[0], xh, qq[0]); qq[0] = MFMA16(n2[0], xl, qq[0]); qq[0] = MFMA16(n2[1], xh, qq[0]);
                    sat_probe(sat, qq[0][0]);
                    if (valid) st4(cen + 512 + 4 * fg, qq[0]);
                }
            } else {           // ppm: 64 -> 32, no bias, xyz component role - 1             (:148, :152)
                f16x8 wp[2][4];
                f16x8 gw[8][2];          // G[c] fragments, both halves
                const float* Lgc = W + lwp.h_gc + lane_n * 4;
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int kgp = 0; kgp < 2; ++kgp) { wp[kgp][2 * m] = LDW(fb + 4096 + ((m * 2 + kgp) * 2) * 256); wp[kgp][2 * m + 1] = LDW(fb + 4096 + ((m * 2 + kgp) * 2 + 1) * 256); }
#pragma unroll
                for (int m = 0; m < 2; ++m) st[m] = ld4(p_state + (size_t)ci * 96 + (role - 1) * 32 + 16 * m + 4 * fg);
#ifdef PESTO_NODEW_PREFETCH_G      // (developer: the G fragments requested while the wave waits for the staged rows - 62 ... 99 registers spilled)
                if (prep) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) { gw[j][0] = LDW(Lgc + (j * 2) * 256); gw[j][1] = LDW(Lgc + (j * 2 + 1) * 256); }
                }
#endif
                rows();
                f32x4 h[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
                PESTO_FIN_MFMA(h, wp[0], zh[0], zl[0])
                PESTO_FIN_MFMA(h, wp[1], zh[1], zl[1])
#pragma unroll
                for (int m = 0; m < 2; ++m) st[m] += h[m];
                finish_state();
                if (prep) {
                    post(st);
                    {   // G[c] blocks 0..7 straight from the own slice p[c], c = role - 1
                        f16x8 ph, pl;
                        split8(st[0], st[1], ph, pl);
#pragma unroll
                        for (int half = 0; half < 2; ++half) {
                            f32x4 a[4];
#ifndef PESTO_NODEW_PREFETCH_G
#pragma unroll
                            for (int j = 0; j < 4; ++j) { gw[4 * half + j][0] = LDW(Lgc + ((4 * half + j) * 2) * 256); gw[4 * half + j][1] = LDW(Lgc + ((4 * half + j) * 2 + 1) * 256); }
#endif
#pragma unroll
                            for (int j = 0; j < 4; ++j) a[j] = f32x4{0, 0, 0, 0};
#pragma unroll
                            for (int j = 0; j < 4; ++j) a[j] = MFMA16(gw[4 * half + j][0], ph, a[j]);
#pragma unroll
                            for (int j = 0; j < 4; ++j) a[j] = MFMA16(gw[4 * half + j][0], pl, a[j]);
#pragma unroll
                            for (int j = 0; j < 4; ++j) a[j] = MFMA16(gw[4 * half + j][1], ph, a[j]);
                            sat_probe(sat, a[0][0]);
                            if (valid) {
#pragma unroll
                                for (int j = 0; j < 4; ++j) st4_finite(cen + (4 * half + j) * 64 + (role - 1) * 16 + 4 * fg, a[j]);
                            }
#ifdef PESTO_NODEW_PREFETCH_G
                            // behind the first half of G: the second half's products and the wait for the slices cover the round trip
                            if (half == 0) { __builtin_amdgcn_sched_barrier(0); load_ua(); __builtin_amdgcn_sched_barrier(0); }
#endif
                        }
                    }
#ifndef PESTO_NODEW_PREFETCH_G
                    load_ua();
#endif
                    tile_inputs();
                    ua_products();
                }
            }
#undef PESTO_FIN_MFMA
            if (valid) sat_flush_at(sat, flags, ci);      // (the probes of a node wave are MFMA column fe = centre ci)
            sat = 0.0f;
#ifdef PESTO_NODEW_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
          }      // ntile
#undef LDW
            ++fin_iter;
        }
        return;      // (every probe of a node wave has been flushed with its tile)
      }
    }
    int last_it = 0;      // (TAILR: start of the last iteration)
    for (int it_start = xcd * chunk; it_start < w_end; it_start += nbx * NE * SUBS) {
      last_it = it_start;
      // the lane-derived values of the work loop (indices, LDS addresses, masks) are re-derived per iteration from an opaque copy of the
      // thread index: as loop invariants they would stay in registers across the finish / prepare phase below, which needs them for
      // weight fragments (forty registers; the fragments of that phase otherwise spill in front of its rendezvous)
      int tid_i = threadIdx.x;
      // (-DPESTO_HOIST_NW, measured and dropped: in node-wave mode the item waves have no finish phase, so only the node waves' lane index
      // needs to be opaque; the nn = 8 item block shrinks by 54 of 894 instructions, the kernel needs all 168 registers, reloads two values
      // from scratch per tile - and is 2 % SLOWER, 57.9 -> 59.2 us per launch: profiles/r05_nw16_ab.txt)
#ifdef PESTO_HOIST_NW
      if (FIN && !NODEW) asm volatile("" : "+v"(tid_i));
#else
      if (FIN) asm volatile("" : "+v"(tid_i));
#endif
      const int lane = tid_i & 63, wave = tid_i >> 6;
      const int e = lane & 15, g = lane >> 4;
      // wave priority by age class (waves 0-3 / 4-7 / 8-11 of a workgroup = oldest / middle / youngest wave of their SIMD). At equal
      // priority the arbiter prefers the oldest wave: in a full nn = 64 launch wave 0 reaches the rendezvous of its twelve-wave workgroup
      // 20 - 28 thousand cycles (~25 % of an iteration) ahead of the slowest wave and its SIMD runs on two waves for that long. The
      // priority of a wave's MFMA bursts therefore grows with its youth (1 / 2 / 3; outside the bursts 0 for everybody, as before):
      // nn = 64 -1.2 ... -2.5 % per launch on three boxes (profiles/r05_prio_age_ab.txt); the reverse order (3, 2, 1), a raised base level
      // for the young waves and the same table on the one-tile items' two-pass path (nn = 8 / 16) gain nothing. Scheduling only: same bits.
      // -DPESTO_PRIO_HI_TAB=1,1,1 builds the uniform priorities of rounds 1 - 5.
#ifndef PESTO_PRIO_HI_TAB
#define PESTO_PRIO_HI_TAB 1, 2, 3
#endif
#ifndef PESTO_PRIO_LO_TAB
#define PESTO_PRIO_LO_TAB 0, 0, 0
#endif
      const int wave_p = __builtin_amdgcn_readfirstlane(wave) >> 2;
      (void)wave_p;
#define PESTO_PRIO_HI() prio_by_age<PESTO_PRIO_HI_TAB>(wave_p)
#define PESTO_PRIO_LO() prio_by_age<PESTO_PRIO_LO_TAB>(wave_p)
      // (the two-pass path: one-tile items of nn = 8 / 16, the exact kernels)
#ifndef PESTO_PRIO_HI2_TAB
#define PESTO_PRIO_HI2_TAB 1, 1, 1
#endif
#ifndef PESTO_PRIO_LO2_TAB
#define PESTO_PRIO_LO2_TAB 0, 0, 0
#endif
#define PESTO_PRIO_HI2() prio_by_age<PESTO_PRIO_HI2_TAB>(wave_p)
#define PESTO_PRIO_LO2() prio_by_age<PESTO_PRIO_LO2_TAB>(wave_p)
      const int wslot = (NODEW && wave >= NE) ? 0 : wave;      // (node waves never touch the per-wave scratch)
      auto& ws = sm.ws[wslot];
      float (*zrow)[256] = sm.zrows[wslot][NODEW ? (fin_iter & 1) : 0];
      const bool tail = w_end - it_start < nbx * NE * SUBS;
      const int sstride = tail ? nbx * NE : NE;
      const int base = it_start + jb * NE * (tail ? 1 : SUBS);
      // node-wave mode: this generation of staging rows was last used two iterations ago - the node waves must have read it
      if (NODEW && wave < NE && fin_iter >= 2) {
          if (NTEAM == 2) lds_wait_ge(&sm.xflag[XF_CONSUMED + (fin_iter & 1)], 4 * (fin_iter >> 1));      // (one counter per generation = per team)
          else lds_wait_ge(&sm.xflag[XF_CONSUMED], 4 * NWT * (fin_iter - 1));
      }
#pragma unroll 1
      for (int sub = 0; sub < SUBS; ++sub) {
      const int work = base + sub * sstride + wave;
      bool item_on = work < w_end && (!NODEW || wave < NE);
      if constexpr (!F16 && !FIN) {      // the exact kernels as AUTO's fp32 repeat: items without a centre of a flagged structure are skipped
          if (item_on && only_fl) item_on = rows_flagged(flags, work * A, A, N1, lane);
      }
      if (item_on) {
        const int c0 = work * A;
        if constexpr (M32) {
            TRACE32(19);
            {   // rows of this work item: lane = row
                const int a = lane / NN, c = lane % NN, i = c0 + a;
                const bool valid = i < N1 && lane < 16 * TI;
                const size_t src = (size_t)min(i, N1 - 1) * KMAX + c;
                const int nbv = ids_s[src];
                const float4 gg = geo[src];
                ws.nb[lane] = valid ? nbv : 0;
                ws.geo[0][lane] = valid ? gg.x : 0.f; ws.geo[1][lane] = valid ? gg.y : 0.f; ws.geo[2][lane] = valid ? gg.z : 0.f;
                ws.geo[3][lane] = valid ? gg.w : 0.f;
            }
            __builtin_amdgcn_wave_barrier();
            PHASE_INIT();
            TRACE32(20);
            edge_item32<NN, TI / 2>(ws, sm.w, zrow, sub, lane, c0, N1, make_rsrc(rec_nb), make_rsrc(rec_cen), make_rsrc(p_state), 0.69314718055994530942f /* Q' = Q log2(e) / sdk: ln 2 gives logit / sdk back */, sat, tr_n);
            // (developer variant: a lane's probes cover every centre of the item - all of them are flagged)
#pragma unroll
            for (int a = 0; a < A; ++a) if (c0 + a < N1) sat_flush_at(sat, flags, c0 + a);
            sat = 0.0f;
            PHASE_MARK(1);
            if (!FIN) {      // unfused variant: the Z rows of the item's centres go to memory, the node kernel applies the output MLPs
#pragma unroll
                for (int a = 0; a < A; ++a) {
                    const int i = c0 + a;
                    if (i < N1) {
                        float* zo = Z + (size_t)i * REC_Z;
#pragma unroll
                        for (int k = 0; k < 4; ++k) zo[lane + 64 * k] = zrow[a][lane + 64 * k];
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        } else {
        PHASE_INIT();
        if (!(it_start == xcd * chunk && sub == 0)) {   // rows of this work item: lane = row (the first item's rows are staged already)
            const int a = lane / NN, c = lane % NN, i = c0 + a;
            const bool valid = i < N1 && lane < 16 * TI;
            const size_t src = (size_t)min(i, N1 - 1) * KMAX + c;           // unconditional loads, select afterwards
            const int nbv = ids_s[src];
            const float4 gg = geo[src];
            if (WROWS == 64 || lane < WROWS) {
                ws.nb[lane] = valid ? nbv : 0;
                ws.geo[0][lane] = valid ? gg.x : 0.f; ws.geo[1][lane] = valid ? gg.y : 0.f; ws.geo[2][lane] = valid ? gg.z : 0.f;
                ws.geo[3][lane] = valid ? gg.w : 0.f; ws.geo[4][lane] = 1.0f;
            }
        }
        __builtin_amdgcn_wave_barrier();
        PHASE_MARK(0);

        // ------------------------------------------------------------------ pass 1: keys -> logits (eqkm, epkm)
        constexpr bool ONEP = TI == 1 && !PF && HY;      // one-tile items: see l1_issue_ac
        // STASH (round 5, measured and NOT shipped; -DPESTO_STASH_OPERAND builds it): items of several tiles cannot keep the first pass's
        // p_j . r_hat operand (hi, lo: eight registers per tile) until the second pass, and the LDS is full - the second pass gathers the
        // six p_j pieces again, recomputes the projection, splits it and moves it to the MFMA lanes (27 packed VALU + 12 split + 8
        // ds_bpermute + 6 gathers per tile). The variant parks the finished operand in a per-wave slot of global memory (2 KB per tile, two
        // coalesced 16-byte stores per lane), reads it back in the second pass (an L2 hit) and takes the one-tile items' path. Same bits,
        // ~70 issue units fewer per tile - and 2.5 - 5.7 % SLOWER on every layer kernel of the forward, including the nn = 8 one that
        // does not use it (profiles/r05_stash_ab.txt): +390 MB of fabric traffic per nn = 64 launch (233 MB before) costs package power,
        // and the chip sits at its 1,400 W limit - the clock pays for the bytes.
#ifdef PESTO_STASH_OPERAND
        constexpr bool STASH = TI > 1 && !PF && HY && FIN && !M32 && !(NN == 32 && TI == 2);      // (that instantiation spilled 428 B per lane)
#else
        constexpr bool STASH = false;
#endif
        float* const stash = STASH ? Z + ((size_t)(blockIdx.x * WPB + wave) * TI) * 512 + lane * 4 : nullptr;      // [tile][hi | lo][lane][16 B]
        (void)stash;
        // W3SPLIT: the part-3 attention weights (the p_j sums' weights) are kept parity-split in LDS and read 16 bytes at a time. Not in the
        // fine-item nn = 32 instantiation (one-structure launches): at the 168-register limit the two live float4 spill there (20 B per
        // lane, +0.7 us per launch, measured in both pairs of profiles/r04_epilogue_ab.txt). Writer and reader share this switch.
        constexpr bool W3SPLIT = !(FIN && NN == 32 && TI == 2);
        // SP (round 5): ONE pass per tile for the centres of several tiles (nn = 32 / 64). The softmax of the split path is unnormalised
        // since this round (exp2 without the row maximum), so a tile's attention weights exp2(t) are known as soon as its key networks have
        // run: the value network follows in the same tile and reuses its p_j . r_hat operand (what one-tile items do, ONEP) - no second
        // gather of the six p_j pieces, projection, split and lane move per tile, no logits round trip through LDS - and the weighted
        // sums are taken with the unnormalised weights; the row sums are accumulated per lane and 1 / sum is applied once per centre in
        // the epilogue. Gives up the gather prefetch across the tiles of an item. Every instantiation of an nn switches together (the
        // normalisation order changes the rounding: a structure must give the same bits alone and in a batch).
        // Same box (profiles/r05_sp_ab.txt): nn = 64 265.4 -> 253.7 us per launch (-4.4 %), nn = 32 150.0 -> 144.1 (-3.9 %), 1,755 -> 1,818
        // structures/s (+3.6 %). -DPESTO_NO_SP builds the two-pass form of rounds 1 - 4.
#ifndef PESTO_SP_MIN_NN
#define PESTO_SP_MIN_NN 32
#endif
        constexpr int SP_MIN_NN = PESTO_SP_MIN_NN;
#ifndef PESTO_NO_SP
        constexpr bool SP = F16 && HY && FIN && !PF && !M32 && NN >= SP_MIN_NN;
#else
        constexpr bool SP = false;
#endif
        L1RawAC rac2;
        f16x8 pr_h, pr_l;
        (void)rac2; (void)pr_h; (void)pr_l;
        // VFIRST (round 5, measured and NOT shipped; -DPESTO_VFIRST builds it): one-tile items run the VALUE network right behind the key
        // networks, in front of the softmax - the two MLP chains are independent (only the weighted sums need the attention weights), and
        // a one-tile item is a single dependent chain per wave: with two item waves per SIMD the nn = 8 layer runs at 1.58x its issue
        // floor (profiles/r05_issue_floor.md). Same operands and MFMA order per accumulator (same bits) - and no gain: nn = 8 58.9 ->
        // 59.7 us, one structure 0.932 -> 0.939 ms on one box (profiles/r05_vfirst_ab.txt). The compiler keeps the two chains one after the
        // other; moving a chain in front of the softmax does not shorten anything by itself.
#ifdef PESTO_VFIRST
        constexpr bool VFIRST = ONEP && F16 && FIN;
#else
        constexpr bool VFIRST = false;
#endif
        f32x4 v_first[4];
        f32x4 pv_first[4];
        (void)v_first; (void)pv_first;
        {
            // layers 2/3 of the key networks for one tile, raw logits parked in the (not yet used) attention-weight table
            auto keys_of_tile = [&](int t, const f32x4* h1, float* lg_regs = nullptr) {      // lg_regs: the two logits of this lane stay in registers (SP)
                f32x4 acc2[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) acc2[m] = ld4(sm.w + EL_B2 + 16 * m + 4 * g);
                f32x4 kacc = ld4(sm.w + EL_BK + 4 * g);
                if (F16) {   // key networks on f16-split MFMA: eq (h1 blocks 0,1) and ep (blocks 2,3), K = 32 each
                    f16x8 xh[2], xl[2], wh[4], wl[4];
                    split8(h1[0], h1[1], xh[0], xl[0]);
                    split8(h1[2], h1[3], xh[1], xl[1]);
#pragma unroll
                    for (int a = 0; a < 4; ++a) {      // a = net * 2 + ml; fragment order [net][ml][hi|lo]
                        const float* fr = w2f + (size_t)(a * 2) * 256 + lane * 4;
                        wh[a] = ld8h(fr); wl[a] = PESTO_WL(fr);
                    }
#pragma unroll
                    for (int a = 0; a < 4; ++a) acc2[a] = MFMA16(wh[a], xh[a >> 1], acc2[a]);
#pragma unroll
                    for (int a = 0; a < 4; ++a) acc2[a] = MFMA16(wh[a], xl[a >> 1], acc2[a]);
#pragma unroll
                    for (int a = 0; a < 4; ++a) acc2[a] = MFMA16(wl[a], xh[a >> 1], acc2[a]);
                    sat_probe(sat, acc2[0][0]);      // h1 of the scalar-key net (blocks 0, 1) beyond the f16 range
                    sat_probe(sat, acc2[2][0]);      // ... of the vector-key net (blocks 2, 3)
                    // keys: K = 64 = k-group 0 (eq h2 blocks) + k-group 1 (ep h2 blocks); two accumulators, summed
                    f32x4 kacb = f32x4{0, 0, 0, 0};
                    f16x8 kh[2], kl[2];
                    split8(elu4s(acc2[0]), elu4s(acc2[1]), xh[0], xl[0]);
                    split8(elu4s(acc2[2]), elu4s(acc2[3]), xh[1], xl[1]);
#pragma unroll
                    for (int kgp = 0; kgp < 2; ++kgp) {
                        const float* fr = w3k + (size_t)(kgp * 2) * 256 + lane * 4;
                        kh[kgp] = ld8h(fr); kl[kgp] = PESTO_WL(fr);
                    }
                    kacc = MFMA16(kh[0], xh[0], kacc); kacb = MFMA16(kh[1], xh[1], kacb);
                    kacc = MFMA16(kh[0], xl[0], kacc); kacb = MFMA16(kh[1], xl[1], kacb);
                    kacc = MFMA16(kl[0], xh[0], kacc); kacb = MFMA16(kl[1], xh[1], kacb);
                    kacc += kacb;
                } else {
#pragma unroll
                    for (int fbl = 0; fbl < 2; ++fbl) {
                        // eq block fbl -> acc2[0..1], ep block 2+fbl -> acc2[2..3]: four independent chains interleaved
                        f32x4 w[4];
#pragma unroll
                        for (int ml = 0; ml < 2; ++ml) {
                            w[ml] = ld4(w2f + ((size_t)(ml * 2 + fbl) * 64 + lane) * 4);
                            w[2 + ml] = ld4(w2f + 4 * 256 + ((size_t)(ml * 2 + fbl) * 64 + lane) * 4);
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            acc2[0] = MFMA(w[0][r], h1[fbl][r], acc2[0]);
                            acc2[2] = MFMA(w[2][r], h1[2 + fbl][r], acc2[2]);
                            acc2[1] = MFMA(w[1][r], h1[fbl][r], acc2[1]);
                            acc2[3] = MFMA(w[3][r], h1[2 + fbl][r], acc2[3]);
                        }
                    }
                    f32x4 kacb = f32x4{0, 0, 0, 0};
                    f32x4 h2k[4], wk[4];
#pragma unroll
                    for (int m = 0; m < 4; ++m) { h2k[m] = elu4(acc2[m]); wk[m] = ld4(w3k + ((size_t)m * 64 + lane) * 4); }
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            kacc = MFMA(wk[m][r], h2k[m][r], kacc);
                            kacb = MFMA(wk[2 + m][r], h2k[2 + m][r], kacb);
                        }
                    kacc += kacb;
                }
                // lane (e, g): kacc[0..2] = key of part g for edge row; logits against Q[0] (scalar) or Q[1] (vector)
                const int aMine = NN == 8 ? 2 * t + (e >> 3) : (16 * t) / NN;
                const float* Qv = rec_cen + (size_t)ABL_CEN(min(c0 + aMine, N1 - 1)) * REC_CEN + 512 + (g == 0 ? 0 : 6);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float lgt = F16 ? Qv[3 * h] * kacc[0] + Qv[3 * h + 1] * kacc[1] + Qv[3 * h + 2] * kacc[2]      // t = log2(e) logit / sdk: Q' carries the scale
                                          : (Qv[3 * h] * kacc[0] + Qv[3 * h + 1] * kacc[1] + Qv[3 * h + 2] * kacc[2]) * inv_sdk;
                    if (lg_regs) lg_regs[h] = lgt; else ws.wts[h * 4 + g][16 * t + e] = lgt;
                }
            };
            if constexpr (SP) {
                float zq[2][2], zp1[2][3][2];
                f32x4 z3a[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
                float srow[2] = {0.f, 0.f};                          // per lane: sum over the centre's tiles of exp2(t) of (part g, edge e)
                float pi3[3] = {0.f, 0.f, 0.f};
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int k = 0; k < 2; ++k) { zq[h][k] = 0.f; zp1[h][0][k] = zp1[h][1][k] = zp1[h][2][k] = 0.f; }
                const int esub = lane >> 5, quad = (lane & 31) < 24 ? (lane & 31) : (lane & 31) - 24;      // (EPI2 lane layout of the p_j gathers)
#pragma unroll 1
                for (int t = 0; t < TI; ++t) {
                    const TileCtx tcc = tile_ctx<NN, HY>(t, e, g, c0, N1, ws, rec_nb, rec_cen);
                    if (t % TPC == 0) {
                        const int ic = ABL_CEN(min(c0 + (16 * t) / NN, N1 - 1));
#pragma unroll
                        for (int c = 0; c < 3; ++c) pi3[c] = p_state[(size_t)ic * 96 + c * 32 + 16 * (g & 1) + e];
                    }
                    const L1Raw raw = l1_issue<NN>(0, t, lane, tcc, ws, p_state);
                    __builtin_amdgcn_sched_barrier(0);
                    L1Head hd = l1_head<NN>(raw, t, lane, tcc, ws);
                    __builtin_amdgcn_sched_barrier(0);
                    const L1RawAC rac = l1_issue_ac<NN>(4, lane, tcc);      // the second half's A_j chunks / centre columns: in flight under the key networks
                    const f16x8 keep_h = hd.fh, keep_l = hd.fl;
                    __builtin_amdgcn_sched_barrier(0);
                    float lgt[2];
                    {
                        f32x4 h1[4];
                        PESTO_PRIO_HI();
                        l1_tail(hd, 0, lane, g, sm.w + EL_W1P, sm.w + EL_WD, h1, sat);
                        keys_of_tile(t, h1, lgt);
                        PESTO_PRIO_LO();
                    }
                    // unnormalised attention weights of this tile (same table layout as the two-pass code), row sums per lane
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float ex = __builtin_amdgcn_exp2f(lgt[h]);
                        srow[h] += ex;
                        ws.wts[h * 4 + g][16 * t + ((W3SPLIT && g == 3) ? ((e & 1) << 3) + (e >> 1) : e)] = ex;
                    }
                    f32x4 pv[4];
                    {   // neighbours' p_j of the first half of the tile's edges (part-3 sums)
                        int nbj[4];
#pragma unroll
                        for (int i2 = 0; i2 < 4; ++i2) nbj[i2] = ABL_NB(ws.nb[16 * t + 2 * i2 + (esub & 1)]);
#pragma unroll
                        for (int i2 = 0; i2 < 4; ++i2) pv[i2] = ld4(p_state + (size_t)nbj[i2] * 96 + 4 * quad);
                    }
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    // ---- value network of the same tile: the operand (keep_h, keep_l) is the first pass's
                    f32x4 h1v[4];
                    {
                        L1Head hv = l1_head_ac<NN>(rac, keep_h, keep_l, lane, tcc);
                        PESTO_PRIO_HI();
                        l1_tail(hv, 4, lane, g, sm.w + EL_W1P, sm.w + EL_WD, h1v, sat);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (W3SPLIT) {
                        const f32x4 w0 = ld4(&ws.wts[3][16 * t + 8 * (esub & 1)]), w1 = ld4(&ws.wts[7][16 * t + 8 * (esub & 1)]);
#pragma unroll
                        for (int i2 = 0; i2 < 4; ++i2) { z3a[0] += w0[i2] * pv[i2]; z3a[1] += w1[i2] * pv[i2]; }
                    } else {
#pragma unroll
                        for (int i2 = 0; i2 < 4; ++i2) {
                            const int ee = 2 * i2 + (esub & 1);
                            const float w0 = ws.wts[3][16 * t + ee], w1 = ws.wts[7][16 * t + ee];
                            z3a[0] += w0 * pv[i2]; z3a[1] += w1 * pv[i2];
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    f32x4 acc2[4];
#pragma unroll
                    for (int ml = 0; ml < 4; ++ml) acc2[ml] = ld4(sm.w + EL_B2 + 64 + 16 * ml + 4 * g);
#pragma unroll
                    for (int kgp = 0; kgp < 2; ++kgp) {
                        f16x8 xh, xl;
                        split8(h1v[2 * kgp], h1v[2 * kgp + 1], xh, xl);
#pragma unroll
                        for (int m0 = 0; m0 < 4; m0 += 2) {
                            f16x8 wh[2], wl[2];
#pragma unroll
                            for (int ml = 0; ml < 2; ++ml) {
                                const float* fr = w2f + 8 * 256 + (size_t)(((m0 + ml) * 2 + kgp) * 2) * 256 + lane * 4;
                                wh[ml] = ld8h(fr); wl[ml] = PESTO_WL(fr);
                            }
#pragma unroll
                            for (int ml = 0; ml < 2; ++ml) acc2[m0 + ml] = MFMA16(wh[ml], xh, acc2[m0 + ml]);
#pragma unroll
                            for (int ml = 0; ml < 2; ++ml) acc2[m0 + ml] = MFMA16(wh[ml], xl, acc2[m0 + ml]);
#pragma unroll
                            for (int ml = 0; ml < 2; ++ml) acc2[m0 + ml] = MFMA16(wl[ml], xh, acc2[m0 + ml]);
                        }
                    }
                    sat_probe(sat, acc2[0][0]);
                    __builtin_amdgcn_sched_barrier(0);
                    {   // second half of the tile's edges: these loads land during the MFMA phase
                        int nbj[4];
#pragma unroll
                        for (int i2 = 0; i2 < 4; ++i2) nbj[i2] = ABL_NB(ws.nb[16 * t + 8 + 2 * i2 + (esub & 1)]);
#pragma unroll
                        for (int i2 = 0; i2 < 4; ++i2) pv[i2] = ld4(p_state + (size_t)nbj[i2] * 96 + 4 * quad);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    f32x4 h2[4];
#pragma unroll
                    for (int ml = 0; ml < 4; ++ml) h2[ml] = elu4s(acc2[ml]);
                    f32x4 v[4];
#pragma unroll
                    for (int fo = 0; fo < 4; ++fo) v[fo] = ld4(&sm.b3v4[4 * (16 * fo + e)]);
#pragma unroll
                    for (int kgp = 0; kgp < 2; ++kgp) {
                        f16x8 ah, al;
                        split8(h2[2 * kgp], h2[2 * kgp + 1], ah, al);
#pragma unroll
                        for (int f0 = 0; f0 < 4; f0 += 2) {
                            f16x8 bh[2], bl[2];
#pragma unroll
                            for (int fo = 0; fo < 2; ++fo) {
                                const float* fr = w3v + (size_t)(((f0 + fo) * 2 + kgp) * 2) * 256 + lane * 4;
                                bh[fo] = ld8h(fr); bl[fo] = PESTO_WL(fr);
                            }
#pragma unroll
                            for (int fo = 0; fo < 2; ++fo) v[f0 + fo] = MFMA16(ah, bh[fo], v[f0 + fo]);
#pragma unroll
                            for (int fo = 0; fo < 2; ++fo) v[f0 + fo] = MFMA16(al, bh[fo], v[f0 + fo]);
#pragma unroll
                            for (int fo = 0; fo < 2; ++fo) v[f0 + fo] = MFMA16(ah, bl[fo], v[f0 + fo]);
                        }
                    }
                    PESTO_PRIO_LO();
                    if constexpr (W3SPLIT) {
                        const f32x4 w0 = ld4(&ws.wts[3][16 * t + 8 * (esub & 1) + 4]), w1 = ld4(&ws.wts[7][16 * t + 8 * (esub & 1) + 4]);
#pragma unroll
                        for (int i2 = 0; i2 < 4; ++i2) { z3a[0] += w0[i2] * pv[i2]; z3a[1] += w1[i2] * pv[i2]; }
                    } else {
#pragma unroll
                        for (int i2 = 0; i2 < 4; ++i2) {
                            const int ee = 8 + 2 * i2 + (esub & 1);
                            const float w0 = ws.wts[3][16 * t + ee], w1 = ws.wts[7][16 * t + ee];
                            z3a[0] += w0 * pv[i2]; z3a[1] += w1 * pv[i2];
                        }
                    }
                    // attention-weighted sums over this lane's four edges, unnormalised weights (:143-144, first block of Vp :132)
                    const int r0 = 16 * t + 4 * g;
                    const f32x4 gx = ld4(&ws.geo[0][r0]), gy = ld4(&ws.geo[1][r0]), gz = ld4(&ws.geo[2][r0]);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const f32x4 wq = ld4(&ws.wts[h * 4 + 0][r0]), w1 = ld4(&ws.wts[h * 4 + 1][r0]);
                        const f32x4 wx4 = w1 * gx, wy4 = w1 * gy, wz4 = w1 * gz;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            zq[h][0] += wq[r] * v[0][r];
                            zq[h][1] += wq[r] * v[1][r];
                            const float wx = wx4[r], wy = wy4[r], wz = wz4[r];
                            zp1[h][0][0] += wx * v[2][r]; zp1[h][0][1] += wx * v[3][r];
                            zp1[h][1][0] += wy * v[2][r]; zp1[h][1][1] += wy * v[3][r];
                            zp1[h][2][0] += wz * v[2][r]; zp1[h][2][1] += wz * v[3][r];
                        }
                    }
                    if ((t + 1) % TPC != 0) continue;      // the centre continues in the next tile
                    // ---- centre complete: the softmax denominators, then the epilogue (reduce-scatter on permlane swaps) with 1 / sum applied
                    __builtin_amdgcn_sched_barrier(0);
                    float rq[2], rv[2], wsm[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float s_row = row_reduce<true, false>(srow[h]);                    // this part's sum over the centre's edges
                        const float tot_v = xrow<false>(xhalf<false>(g == 0 ? 0.0f : s_row));    // parts 1..3 together, every lane
                        const float tot_q = lane_bcast(s_row, 0), s2 = lane_bcast(s_row, 32);    // part 0 / part 2 (wave-uniform)
                        rq[h] = __builtin_amdgcn_rcpf(tot_q);
                        rv[h] = __builtin_amdgcn_rcpf(tot_v);
                        sat_probe(sat, tot_q); sat_probe(sat, rq[h] * 0x1p27f);                  // range guard of the unsubtracted softmax
                        sat_probe(sat, tot_v); sat_probe(sat, rv[h] * 0x1p27f);
                        wsm[h] = s2 * rv[h];                                                     // the centre's part-2 share (multiplies p_i)
                        srow[h] = 0.f;
                    }
                    if (sat != sat) {
                        const int rowc = c0 + (16 * t) / NN;
                        if (rowc < N1) sat_flush_at(sat, flags, rowc);
                    }
                    sat = 0.0f;
                    float Qt[2], Pt[2][3];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        Qt[h] = swap_add_rows(zq[h][0], zq[h][1]);
#pragma unroll
                        for (int c = 0; c < 3; ++c) Pt[h][c] = swap_add_rows(zp1[h][c][0], zp1[h][c][1]);
                    }
                    Qt[0] = swap_add_halves(Qt[0], Qt[1]);
#pragma unroll
                    for (int c = 0; c < 3; ++c) Pt[0][c] = swap_add_halves(Pt[0][c], Pt[1][c]);
                    {
                        f32x4 za;
#pragma unroll
                        for (int j = 0; j < 4; ++j) za[j] = swap_add_halves(z3a[0][j], z3a[1][j]);
                        if ((lane & 31) < 24) st4(&ws.z3buf[0][lane >> 5][4 * (lane & 31)], za);
                        z3a[0] = f32x4{0, 0, 0, 0}; z3a[1] = f32x4{0, 0, 0, 0};
                    }
                    __builtin_amdgcn_wave_barrier();
                    {
                        const int s_l = 16 * (g & 1) + e, hh = g >> 1;
                        const int slot0 = SUBS > 1 ? sub : (16 * t) / NN;
                        float* zb = zrow[slot0];
                        const float rqh = hh ? rq[1] : rq[0], rvh = hh ? rv[1] : rv[0], wsh = hh ? wsm[1] : wsm[0];
                        zb[hh * 32 + s_l] = Qt[0] * rqh;
#pragma unroll
                        for (int c = 0; c < 3; ++c) zb[64 + c * 64 + hh * 32 + s_l] = (Pt[0][c] + ws.z3buf[0][hh][c * 32 + s_l]) * rvh + wsh * pi3[c];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int k = 0; k < 2; ++k) { zq[h][k] = 0.f; zp1[h][0][k] = zp1[h][1][k] = zp1[h][2][k] = 0.f; }
                    __builtin_amdgcn_wave_barrier();
                }
            } else if (PF) {
                // tile-batched: the four first-layer blocks of a tile are computed together (VALU phase, independent
                // chains), then the layer-2/3 MFMA chains run dense; the NEXT tile's gathers are issued in between
                TileCtx tc[2];
                tc[0] = tile_ctx<NN>(0, e, g, c0, N1, ws, rec_nb, rec_cen);
                L1Ops ops[2][4];
#pragma unroll
                for (int fb = 0; fb < 4; ++fb) ops[0][fb] = l1_fetch<NN>(fb, lane, g, tc[0].cenA, tc[0].cenB, tc[0].recj);
#pragma unroll
                for (int t = 0; t < TI; ++t) {
                    const TileCtx& tcc = tc[t & 1];
                    f32x4 h1[4];
#pragma unroll
                    for (int fb = 0; fb < 4; ++fb)
                        h1[fb] = l1_compute<NN>(ops[t & 1][fb], fb, g, tcc.bgA, tcc.bgB, sm.w + EL_WD, tcc.d, tcc.rx, tcc.ry, tcc.rz);
                    __builtin_amdgcn_sched_barrier(0);
                    if (t < 3) {
                        tc[(t + 1) & 1] = tile_ctx<NN>(t + 1, e, g, c0, N1, ws, rec_nb, rec_cen);
                        const TileCtx& tn = tc[(t + 1) & 1];
#pragma unroll
                        for (int fb = 0; fb < 4; ++fb) ops[(t + 1) & 1][fb] = l1_fetch<NN>(fb, lane, g, tn.cenA, tn.cenB, tn.recj);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    keys_of_tile(t, h1);
                }
            } else {
                {
                    // software pipeline over the four tiles: the NEXT tile's gathers are issued as soon as this tile's raw loads
                    // have been consumed, and fly during this tile's MFMA / ELU / key-network work
                    TileCtx tcc = tile_ctx<NN, HY>(0, e, g, c0, N1, ws, rec_nb, rec_cen);
                    L1Raw raw = l1_issue<NN>(0, 0, lane, tcc, ws, p_state);
#pragma unroll 1
                    for (int t = 0; t < TI; ++t) {
                        if (SAT2 && t == TI / 2) { const float tmp = sat; sat = sat_b; sat_b = tmp; }      // the second centre's tiles start
                        L1Head hd = l1_head<NN>(raw, t, lane, tcc, ws);
                        __builtin_amdgcn_sched_barrier(0);
                        if (ONEP) { rac2 = l1_issue_ac<NN>(4, lane, tcc); pr_h = hd.fh; pr_l = hd.fl; }
                        if (STASH) {
                            st4(stash + t * 512, __builtin_bit_cast(f32x4, hd.fh));
                            st4(stash + t * 512 + 256, __builtin_bit_cast(f32x4, hd.fl));
                        }
                        if (t < TI - 1) {
                            tcc = tile_ctx<NN, HY>(t + 1, e, g, c0, N1, ws, rec_nb, rec_cen);
                            raw = l1_issue<NN>(0, t + 1, lane, tcc, ws, p_state);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        f32x4 h1[4];
                        // wave priority: a wave inside its MFMA burst (first-layer tail, key / value networks) goes ahead of waves that are in
                        // VALU / LDS phases (softmax, weighted sums, finalize) - measured +3.8 % (levels 1..3 alike)
                        PESTO_PRIO_HI2();
                        l1_tail(hd, 0, lane, g, sm.w + EL_W1P, sm.w + EL_WD, h1, sat);
                        keys_of_tile(t, h1);
                        if constexpr (VFIRST) {      // (t == 0: a one-tile item)
                            L1Head hv = l1_head_ac<NN>(rac2, pr_h, pr_l, lane, tcc);
                            f32x4 h1v[4];
                            l1_tail(hv, 4, lane, g, sm.w + EL_W1P, sm.w + EL_WD, h1v, sat);
                            f32x4 acc2v[4];
#pragma unroll
                            for (int ml = 0; ml < 4; ++ml) acc2v[ml] = ld4(sm.w + EL_B2 + 64 + 16 * ml + 4 * g);
#pragma unroll
                            for (int kgp = 0; kgp < 2; ++kgp) {
                                f16x8 xh, xl;
                                split8(h1v[2 * kgp], h1v[2 * kgp + 1], xh, xl);
#pragma unroll
                                for (int m0 = 0; m0 < 4; m0 += 2) {
                                    f16x8 wh[2], wl[2];
#pragma unroll
                                    for (int ml = 0; ml < 2; ++ml) {
                                        const float* fr = w2f + 8 * 256 + (size_t)(((m0 + ml) * 2 + kgp) * 2) * 256 + lane * 4;
                                        wh[ml] = ld8h(fr); wl[ml] = PESTO_WL(fr);
                                    }
#pragma unroll
                                    for (int ml = 0; ml < 2; ++ml) acc2v[m0 + ml] = MFMA16(wh[ml], xh, acc2v[m0 + ml]);
#pragma unroll
                                    for (int ml = 0; ml < 2; ++ml) acc2v[m0 + ml] = MFMA16(wh[ml], xl, acc2v[m0 + ml]);
#pragma unroll
                                    for (int ml = 0; ml < 2; ++ml) acc2v[m0 + ml] = MFMA16(wl[ml], xh, acc2v[m0 + ml]);
                                }
                            }
                            sat_probe(sat, acc2v[0][0]);          // h1 of the value net beyond the f16 range
                            f32x4 h2v[4];
#pragma unroll
                            for (int ml = 0; ml < 4; ++ml) h2v[ml] = elu4s(acc2v[ml]);
#pragma unroll
                            for (int fo = 0; fo < 4; ++fo) v_first[fo] = ld4(&sm.b3v4[4 * (16 * fo + e)]);
#pragma unroll
                            for (int kgp = 0; kgp < 2; ++kgp) {
                                f16x8 ah, al;
                                split8(h2v[2 * kgp], h2v[2 * kgp + 1], ah, al);
#pragma unroll
                                for (int f0 = 0; f0 < 4; f0 += 2) {
                                    f16x8 bh[2], bl[2];
#pragma unroll
                                    for (int fo = 0; fo < 2; ++fo) {
                                        const float* fr = w3v + (size_t)(((f0 + fo) * 2 + kgp) * 2) * 256 + lane * 4;
                                        bh[fo] = ld8h(fr); bl[fo] = PESTO_WL(fr);
                                    }
#pragma unroll
                                    for (int fo = 0; fo < 2; ++fo) v_first[f0 + fo] = MFMA16(ah, bh[fo], v_first[f0 + fo]);
#pragma unroll
                                    for (int fo = 0; fo < 2; ++fo) v_first[f0 + fo] = MFMA16(al, bh[fo], v_first[f0 + fo]);
#pragma unroll
                                    for (int fo = 0; fo < 2; ++fo) v_first[f0 + fo] = MFMA16(ah, bl[fo], v_first[f0 + fo]);
                                }
                            }
                        }
                        PESTO_PRIO_LO2();
                        if constexpr (VFIRST) {      // the first half of the tile's p_j gathers for the part-3 sums: in flight during the softmax
                            const int esub0 = lane >> 5, quad0 = (lane & 31) < 24 ? (lane & 31) : (lane & 31) - 24;
                            int nbj[4];
#pragma unroll
                            for (int i2 = 0; i2 < 4; ++i2) nbj[i2] = ABL_NB(ws.nb[2 * i2 + (esub0 & 1)]);
#pragma unroll
                            for (int i2 = 0; i2 < 4; ++i2) pv_first[i2] = ld4(p_state + (size_t)nbj[i2] * 96 + 4 * quad0);
                        }
                    }
                    if (SAT2) { const float tmp = sat; sat = sat_b; sat_b = tmp; }      // sat: first centre again, sat_b: second
                }
            }
        }
        if constexpr (!SP) {
        float lg[4][2];
#pragma unroll
        for (int t = 0; t < TI; ++t) { lg[t][0] = ws.wts[g][16 * t + e]; lg[t][1] = ws.wts[4 + g][16 * t + e]; }
        PHASE_MARK(1);
        // ------------------------------------------------------------------ softmax per centre  (:139-140)
        // scalar: over the NN rows of part 0; vector: over the 3*NN slots of parts 1..3 together
        if constexpr (F16) {
            // Split path: w = exp2(t) / sum exp2(t), t = log2(e) logit / sdk (the scale rides on Q', pesto_schema.cpp) - torch's softmax
            // (:139-140) subtracts the row maximum first, which is the same function and only protects the exponent range. Here the
            // range is GUARDED instead of protected: the logits of the trained checkpoints stay within -39 .. +64
            // (profiles/r05_logit_range.txt; fp32 holds e^+-87), and a centre whose sum leaves [2^-101, inf) - overflow, or every
            // term flushed - trips the range guard of its structure (sat_probe), which PESTO_PRECISION_AUTO repeats on the exact fp32
            // kernels (they keep the max-subtracted form below). Per head and centre that removes a 16-lane max reduction, three
            // readlanes and the subtraction; the cross-part sum of the vector softmax is two v_permlane swaps instead of three
            // v_readlane + moves (nn = 8: six, selected per lane half), and the centre's part-2 share is a multiply by the reciprocal
            // the weights use anyway (it was an IEEE division: ten instructions).
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float ex[4], sr[4];
#pragma unroll
                for (int t = 0; t < TI; ++t) { ex[t] = __builtin_amdgcn_exp2f(lg[t][h]); sr[t] = ex[t]; }
                if (TPC == 4) sr[0] = (sr[0] + sr[1]) + (sr[2] + sr[3]);
                if (TPC == 2) {
                    sr[0] = sr[0] + sr[1];
                    if (TI == 4) sr[2] = sr[2] + sr[3];
                }
#pragma unroll
                for (int t = 0; t < TI; t += TPC) {      // one centre per step (nn = 8: the tile's two centres in the two halves of every lane row)
                    const float srow = row_reduce<(NN >= 16), false>(sr[t]);                 // this part's sum over the centre's edges
                    const float tot_v = xrow<false>(xhalf<false>(g == 0 ? 0.0f : srow));     // parts 1..3 together (every lane row gets it)
                    const float tot = g == 0 ? srow : tot_v;
                    const float rinv = __builtin_amdgcn_rcpf(tot);
                    float& sg = (SAT2 && t >= TI / 2) ? sat_b : sat;
                    sat_probe(sg, tot);                     // sum overflowed (some logit > ~88)
                    sat_probe(sg, rinv * 0x1p27f);          // sum below 2^-101: every term underflowed
#pragma unroll
                    for (int tt = t; tt < t + TPC; ++tt)
                        // the part-3 rows (g == 3: the weights of the p_j sums) are stored parity-split within the tile - [edges 0, 2, .. 14 | 1, 3, .. 15] -
                        // because their only reader takes the edges of ONE parity: two 16-byte reads per row and tile instead of eight 4-byte ones
                        ws.wts[h * 4 + g][16 * tt + ((W3SPLIT && g == 3) ? ((e & 1) << 3) + (e >> 1) : e)] = ex[tt] * rinv;
                    // centre-level sum of the part-2 weights (what multiplies p_i in Zp): written by the part-2 lanes
                    if (g == 2 && (NN == 8 ? (e & 7) == 0 : e == 0)) {
                        const int a = NN == 8 ? 2 * t + (e >> 3) : (16 * t) / NN;
                        ws.wsum[a][h] = srow * rinv;
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float mx[4], ex[4], sr[4];
#pragma unroll
                for (int t = 0; t < TI; ++t) mx[t] = lg[t][h];
                if (TPC == 4) { const float m = fmaxf(fmaxf(mx[0], mx[1]), fmaxf(mx[2], mx[3])); mx[0] = mx[1] = mx[2] = mx[3] = m; }
                if (TPC == 2) {
                    const float m0 = fmaxf(mx[0], mx[1]); mx[0] = mx[1] = m0;
                    if (TI == 4) { const float m1 = fmaxf(mx[2], mx[3]); mx[2] = mx[3] = m1; }
                }
#pragma unroll
                for (int t = 0; t < TI; ++t) {
                    const float m = row_reduce<(NN >= 16), true>(mx[t]);
                    // rows 1..3 (the three vector-key chunks) share one softmax: fetch their row results (wave-uniform lanes)
                    float v1, v2, v3;
                    if (NN >= 16) { v1 = lane_bcast(m, 16); v2 = lane_bcast(m, 32); v3 = lane_bcast(m, 48); }
                    else {
                        v1 = e < 8 ? lane_bcast(m, 16) : lane_bcast(m, 24);
                        v2 = e < 8 ? lane_bcast(m, 32) : lane_bcast(m, 40);
                        v3 = e < 8 ? lane_bcast(m, 48) : lane_bcast(m, 56);
                    }
                    mx[t] = g == 0 ? m : fmaxf(v1, fmaxf(v2, v3));
                    ex[t] = __expf(lg[t][h] - mx[t]);
                    sr[t] = ex[t];
                }
                if (TPC == 4) { const float s = (sr[0] + sr[1]) + (sr[2] + sr[3]); sr[0] = sr[1] = sr[2] = sr[3] = s; }
                if (TPC == 2) {
                    const float s0 = sr[0] + sr[1]; sr[0] = sr[1] = s0;
                    if (TI == 4) { const float s1 = sr[2] + sr[3]; sr[2] = sr[3] = s1; }
                }
#pragma unroll
                for (int t = 0; t < TI; ++t) {
                    const float sm_ = row_reduce<(NN >= 16), false>(sr[t]);
                    float s1, s2, s3;
                    if (NN >= 16) { s1 = lane_bcast(sm_, 16); s2 = lane_bcast(sm_, 32); s3 = lane_bcast(sm_, 48); }
                    else {
                        s1 = e < 8 ? lane_bcast(sm_, 16) : lane_bcast(sm_, 24);
                        s2 = e < 8 ? lane_bcast(sm_, 32) : lane_bcast(sm_, 40);
                        s3 = e < 8 ? lane_bcast(sm_, 48) : lane_bcast(sm_, 56);
                    }
                    const float tot = g == 0 ? sm_ : (s1 + s2) + s3;
                    // the part-3 rows (g == 3: the weights of the p_j sums) are stored parity-split within the tile - [edges 0, 2, .. 14 | 1, 3, .. 15] -
                    // because their only reader takes the edges of ONE parity: two 16-byte reads per row and tile instead of eight 4-byte ones
                    ws.wts[h * 4 + g][16 * t + ((W3SPLIT && g == 3) ? ((e & 1) << 3) + (e >> 1) : e)] = ex[t] * __builtin_amdgcn_rcpf(tot);
                    // centre-level sum of the part-2 weights (what multiplies p_i in Zp): written by the part-2 lanes
                    if (g == 2 && (NN == 8 ? (e & 7) == 0 : e == 0) && (t % TPC) == 0) {
                        const int a = NN == 8 ? 2 * t + (e >> 3) : (16 * t) / NN;
                        ws.wsum[a][h] = s2 / ((s1 + s2) + s3);
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        PHASE_MARK(2);

        // ------------------------------------------------------------------ pass 2: values (evm) and the weighted sums
        float zq[2][2], zp1[2][3][2];
        f32x4 z3a[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}}, z3b[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int k = 0; k < 2; ++k) { zq[h][k] = 0.f; zp1[h][0][k] = zp1[h][1][k] = zp1[h][2][k] = 0.f; }

        TileCtx tcn = tile_ctx<NN, HY>(0, e, g, c0, N1, ws, rec_nb, rec_cen);
        L1Ops pre[4];
        if (PF) {
#pragma unroll
            for (int fbl = 0; fbl < 4; ++fbl) pre[fbl] = l1_fetch<NN>(4 + fbl, lane, g, tcn.cenA, tcn.cenB, tcn.recj);
        }
        float pi_pre[2][2] = {{0.f, 0.f}, {0.f, 0.f}};     // centre's own p_i (second block of Vp, :133), fetched a tile phase early
        // EPI2 (the shipped kernels): the centre epilogue as a reduce-scatter over the lane groups on v_permlane swaps - every lane ends with
        // the totals of ITS four Z elements (q and p[0..2] of feature 16 (g & 1) + e, head g >> 1; nn = 8: eight, centre g >> 1, both
        // heads), completes them in registers and stores them once. The ds_bpermute form gave every lane all sixteen totals, parked them
        // in LDS and completed them there (read - modify - write): ~250 instructions per centre against ~80.
        constexpr bool EPI2 = FIN && F16 && !PF;
        float pi3[3] = {0.f, 0.f, 0.f};                     // EPI2: p_i[c][16 (g & 1) + e] of this lane's centre
        (void)pi3;
        for (int t = 0; t < TI; ++t) {
            const TileCtx tc = PF ? tcn : tile_ctx<NN, HY>(t, e, g, c0, N1, ws, rec_nb, rec_cen);
            if (t % TPC == 0) {
                if (EPI2) {
                    const int ic = ABL_CEN(min(c0 + (NN == 8 ? 2 * t + (g >> 1) : (16 * t) / NN), N1 - 1));
#pragma unroll
                    for (int c = 0; c < 3; ++c) pi3[c] = p_state[(size_t)ic * 96 + c * 32 + 16 * (g & 1) + e];
                } else {
#pragma unroll
                for (int sel = 0; sel < (NN == 8 ? 2 : 1); ++sel) {
                    const int ic = ABL_CEN(min(c0 + (NN == 8 ? 2 * t + sel : (16 * t) / NN), N1 - 1));
                    pi_pre[sel][0] = p_state[(size_t)ic * 96 + lane];
                    pi_pre[sel][1] = p_state[(size_t)ic * 96 + 64 + (lane & 31)];
                }
                }
            }
            // neighbours' p_j of this tile (third block of Vp, :134) as 16-byte gathers: lane = (esub = lane / 24, quad =
            // lane % 24) reads floats 4*quad..+3 of the 96-vector of edges 2i + esub; issued first, consumed after the
            // first-layer VALU work below
            // (lanes 48..63 duplicate lanes 0..15's addresses; their sums are never read - no divergent branch around the loads)
            // (EPI2: the two edge parities are the two lane halves - 24 of 32 lanes each carry a piece, the others repeat pieces 0..7 -
            // so that the fold over the parities is a half swap)
            const int esub = EPI2 ? lane >> 5 : lane / 24, quad = EPI2 ? ((lane & 31) < 24 ? (lane & 31) : (lane & 31) - 24) : lane - 24 * esub;
            f32x4 pv[4];
            if constexpr (VFIRST) {
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) pv[i2] = pv_first[i2];
            } else {
                int nbj[4];
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) nbj[i2] = ABL_NB(ws.nb[16 * t + 2 * i2 + (esub & 1)]);
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) pv[i2] = ld4(p_state + (size_t)nbj[i2] * 96 + 4 * quad);
            }
            f32x4 pvB[4];
            (void)pvB;
            if constexpr (VFIRST) {      // second half of the tile's edges, requested at once (no value network to hide them behind)
                int nbj[4];
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) nbj[i2] = ABL_NB(ws.nb[16 * t + 8 + 2 * i2 + (esub & 1)]);
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) pvB[i2] = ld4(p_state + (size_t)nbj[i2] * 96 + 4 * quad);
            }
            f32x4 h1[4];
            if constexpr (VFIRST) {
            } else if (PF) {
#pragma unroll
                for (int fbl = 0; fbl < 4; ++fbl)
                    h1[fbl] = l1_compute<NN>(pre[fbl], 4 + fbl, g, tc.bgA, tc.bgB, sm.w + EL_WD, tc.d, tc.rx, tc.ry, tc.rz);
            } else {
                L1Head hd;
                if (ONEP) {
                    hd = l1_head_ac<NN>(rac2, pr_h, pr_l, lane, tc);
                } else if (STASH) {
                    const L1RawAC rac = l1_issue_ac<NN>(4, lane, tc);
                    const f32x4 sh = ld4(stash + t * 512), sl = ld4(stash + t * 512 + 256);
                    __builtin_amdgcn_sched_barrier(0);
                    hd = l1_head_ac<NN>(rac, __builtin_bit_cast(f16x8, sh), __builtin_bit_cast(f16x8, sl), lane, tc);
                } else {
                    const L1Raw raw = l1_issue<NN>(4, t, lane, tc, ws, p_state);
                    __builtin_amdgcn_sched_barrier(0);
                    hd = l1_head<NN>(raw, t, lane, tc, ws);
                }
                PESTO_PRIO_HI2();
                l1_tail(hd, 4, lane, g, sm.w + EL_W1P, sm.w + EL_WD, h1, sat);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (W3SPLIT) {
                const f32x4 w0 = ld4(&ws.wts[3][16 * t + 8 * (esub & 1)]), w1 = ld4(&ws.wts[7][16 * t + 8 * (esub & 1)]);      // edges 2 i2 + parity, i2 = 0..3
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) { z3a[0] += w0[i2] * pv[i2]; z3a[1] += w1[i2] * pv[i2]; }
            } else {
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) {
                    const int ee = 2 * i2 + (esub & 1);
                    const float w0 = ws.wts[3][16 * t + ee], w1 = ws.wts[7][16 * t + ee];
                    z3a[0] += w0 * pv[i2]; z3a[1] += w1 * pv[i2];
                }
            }
            // the first-layer operands of the NEXT tile fly during this tile's MFMA phase
            if (PF && t < 3) {
                tcn = tile_ctx<NN>(t + 1, e, g, c0, N1, ws, rec_nb, rec_cen);
                if (PF) {
#pragma unroll
                    for (int fbl = 0; fbl < 2; ++fbl) pre[fbl] = l1_fetch<NN>(4 + fbl, lane, g, tcn.cenA, tcn.cenB, tcn.recj);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            f32x4 v[4];
            if constexpr (VFIRST) {
#pragma unroll
                for (int fo = 0; fo < 4; ++fo) v[fo] = v_first[fo];
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) pv[i2] = pvB[i2];
            } else {
            f32x4 acc2[4];
#pragma unroll
            for (int ml = 0; ml < 4; ++ml) acc2[ml] = ld4(sm.w + EL_B2 + 64 + 16 * ml + 4 * g);
            if (F16) {   // value network layer 2 on v_mfma_f32_16x16x32_f16, operands split into f16 hi/lo pairs
#pragma unroll
                for (int kgp = 0; kgp < 2; ++kgp) {
                    f16x8 xh, xl;
                    split8(h1[2 * kgp], h1[2 * kgp + 1], xh, xl);
                    constexpr int G = PF ? 4 : 2;      // output blocks per fragment batch (2 keeps the lean build under 168 VGPRs)
#pragma unroll
                    for (int m0 = 0; m0 < 4; m0 += G) {
                        f16x8 wh[G], wl[G];
#pragma unroll
                        for (int ml = 0; ml < G; ++ml) {
                            const float* fr = w2f + 8 * 256 + (size_t)(((m0 + ml) * 2 + kgp) * 2) * 256 + lane * 4;
                            wh[ml] = ld8h(fr); wl[ml] = PESTO_WL(fr);
                        }
#pragma unroll
                        for (int ml = 0; ml < G; ++ml) acc2[m0 + ml] = MFMA16(wh[ml], xh, acc2[m0 + ml]);
#pragma unroll
                        for (int ml = 0; ml < G; ++ml) acc2[m0 + ml] = MFMA16(wh[ml], xl, acc2[m0 + ml]);
#pragma unroll
                        for (int ml = 0; ml < G; ++ml) acc2[m0 + ml] = MFMA16(wl[ml], xh, acc2[m0 + ml]);
                    }
                }
                sat_probe(sat, acc2[0][0]);          // h1 of the value net beyond the f16 range
            } else {
#pragma unroll
                for (int fbl = 0; fbl < 4; ++fbl) mfma_multi<4, 4>(w2f + 8 * 256, 0, fbl, lane, h1[fbl], acc2);
            }
            PHASE_MARK(3);
            __builtin_amdgcn_sched_barrier(0);
            {   // second half of the tile's edges: these loads land during the MFMA phase
                int nbj[4];
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) nbj[i2] = ABL_NB(ws.nb[16 * t + 8 + 2 * i2 + (esub & 1)]);
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) pv[i2] = ld4(p_state + (size_t)nbj[i2] * 96 + 4 * quad);
            }
            if (PF && t < 3) {   // second half of the next tile's first-layer operands: in flight during the value MFMAs
#pragma unroll
                for (int fbl = 2; fbl < 4; ++fbl) pre[fbl] = l1_fetch<NN>(4 + fbl, lane, g, tcn.cenA, tcn.cenB, tcn.recj);
            }
            __builtin_amdgcn_sched_barrier(0);
            f32x4 h2[4];
#pragma unroll
            for (int ml = 0; ml < 4; ++ml) h2[ml] = F16 ? elu4s(acc2[ml]) : elu4(acc2[ml]);
            // V[edge 16t + 4g + r][feature 16fo + e]: edges as rows (A operand = h2), weights as B operand
#pragma unroll
            for (int fo = 0; fo < 4; ++fo) {
                if constexpr (HY && !M32) {
                    v[fo] = ld4(&sm.b3v4[4 * (16 * fo + e)]);
                } else {
                    const float b = sm.w[EL_B3V + 16 * fo + e];
                    v[fo] = f32x4{b, b, b, b};
                }
            }
            if (F16) {
#pragma unroll
                for (int kgp = 0; kgp < 2; ++kgp) {
                    f16x8 ah, al;
                    split8(h2[2 * kgp], h2[2 * kgp + 1], ah, al);
                    constexpr int G = PF ? 4 : 2;
#pragma unroll
                    for (int f0 = 0; f0 < 4; f0 += G) {
                        f16x8 bh[G], bl[G];
#pragma unroll
                        for (int fo = 0; fo < G; ++fo) {
                            const float* fr = w3v + (size_t)(((f0 + fo) * 2 + kgp) * 2) * 256 + lane * 4;
                            bh[fo] = ld8h(fr); bl[fo] = PESTO_WL(fr);
                        }
#pragma unroll
                        for (int fo = 0; fo < G; ++fo) v[f0 + fo] = MFMA16(ah, bh[fo], v[f0 + fo]);
#pragma unroll
                        for (int fo = 0; fo < G; ++fo) v[f0 + fo] = MFMA16(al, bh[fo], v[f0 + fo]);
#pragma unroll
                        for (int fo = 0; fo < G; ++fo) v[f0 + fo] = MFMA16(ah, bl[fo], v[f0 + fo]);
                    }
                }
            } else {
#pragma unroll
                for (int ml = 0; ml < 4; ++ml) {
                    f32x4 wv[4];
#pragma unroll
                    for (int fo = 0; fo < 4; ++fo) wv[fo] = ld4(w3v + ((size_t)(fo * 4 + ml) * 64 + lane) * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int fo = 0; fo < 4; ++fo) v[fo] = MFMA(h2[ml][r], wv[fo][r], v[fo]);
                }
            }
            }      // !VFIRST
            PHASE_MARK(4);
            PESTO_PRIO_LO2();
            if constexpr (W3SPLIT) {
                const f32x4 w0 = ld4(&ws.wts[3][16 * t + 8 * (esub & 1) + 4]), w1 = ld4(&ws.wts[7][16 * t + 8 * (esub & 1) + 4]);   // edges 8 + 2 i2 + parity
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) {
                    if (NN == 8) { z3b[0] += w0[i2] * pv[i2]; z3b[1] += w1[i2] * pv[i2]; }
                    else { z3a[0] += w0[i2] * pv[i2]; z3a[1] += w1[i2] * pv[i2]; }
                }
            } else {
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) {
                    const int ee = 8 + 2 * i2 + (esub & 1);
                    const float w0 = ws.wts[3][16 * t + ee], w1 = ws.wts[7][16 * t + ee];
                    z3a[0] += w0 * pv[i2]; z3a[1] += w1 * pv[i2];      // (NN == 32 here)
                }
            }
            // attention-weighted sums over this lane's four edges (:143-144, first block of Vp :132)
            const int r0 = 16 * t + 4 * g;
            const f32x4 gx = ld4(&ws.geo[0][r0]), gy = ld4(&ws.geo[1][r0]), gz = ld4(&ws.geo[2][r0]);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 wq = ld4(&ws.wts[h * 4 + 0][r0]), w1 = ld4(&ws.wts[h * 4 + 1][r0]);
                const f32x4 wx4 = w1 * gx, wy4 = w1 * gy, wz4 = w1 * gz;      // (vector form: packed multiplies)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    zq[h][0] += wq[r] * v[0][r];
                    zq[h][1] += wq[r] * v[1][r];
                    const float wx = wx4[r], wy = wy4[r], wz = wz4[r];
                    zp1[h][0][0] += wx * v[2][r]; zp1[h][0][1] += wx * v[3][r];
                    zp1[h][1][0] += wy * v[2][r]; zp1[h][1][1] += wy * v[3][r];
                    zp1[h][2][0] += wz * v[2][r]; zp1[h][2][1] += wz * v[3][r];
                }
            }
            PHASE_MARK(5);
            if ((t + 1) % TPC != 0) continue;   // centre continues in the next tile
            if (F16) {   // range guard of the centre(s) that end here: this lane's probes of both passes were columns (edges) of that centre
                if (sat != sat) {      // (rare: everything it needs is derived inside the branch)
                    const int rowc = c0 + (NN == 8 ? 2 * t + (e >> 3) : (16 * t) / NN);
                    if (rowc < N1) sat_flush_at(sat, flags, rowc);
                }
                // (unfused developer kernels hold more than two centres per item: their probes stay sticky within the item, which can
                // only flag too many of the item's centres, never too few)
                if (FIN) { sat = sat_b; sat_b = 0.0f; }
            }
            if constexpr (EPI2) {
                // ---- centre(s) complete (EPI2): reduce-scatter over the lane groups, finish in registers, one store per element
                __builtin_amdgcn_sched_barrier(0);
                float Qt[2], Pt[2][3];
#pragma unroll
                for (int h = 0; h < 2; ++h) {      // rows (g, g ^ 1): even rows end with the k = 0 totals, odd rows with the k = 1 totals
                    Qt[h] = swap_add_rows(zq[h][0], zq[h][1]);
#pragma unroll
                    for (int c = 0; c < 3; ++c) Pt[h][c] = swap_add_rows(zp1[h][c][0], zp1[h][c][1]);
                }
                if (NN >= 16) {                    // halves: lanes 0..31 end with head 0, lanes 32..63 with head 1 (nn = 8: the halves are two centres)
                    Qt[0] = swap_add_halves(Qt[0], Qt[1]);
#pragma unroll
                    for (int c = 0; c < 3; ++c) Pt[0][c] = swap_add_halves(Pt[0][c], Pt[1][c]);
                }
                {   // p_j sums: fold the two edge parities (lane halves): lanes 0..31 end with head 0, lanes 32..63 with head 1
                    f32x4 za, zb4;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        za[j] = swap_add_halves(z3a[0][j], z3a[1][j]);
                        if (NN == 8) zb4[j] = swap_add_halves(z3b[0][j], z3b[1][j]);
                    }
                    if ((lane & 31) < 24) {
                        st4(&ws.z3buf[0][lane >> 5][4 * (lane & 31)], za);
                        if (NN == 8) st4(&ws.z3buf[1][lane >> 5][4 * (lane & 31)], zb4);
                    }
#pragma unroll
                    for (int h = 0; h < 2; ++h) { z3a[h] = f32x4{0, 0, 0, 0}; z3b[h] = f32x4{0, 0, 0, 0}; }
                }
                __builtin_amdgcn_wave_barrier();
                const int s_l = 16 * (g & 1) + e;                                        // this lane's feature
                if (NN >= 16) {
                    const int slot0 = SUBS > 1 ? sub : (16 * t) / NN;                    // FIN keeps every centre of the iteration staged
                    const int a = (16 * t) / NN, hh = g >> 1;
                    float* zb = zrow[slot0];
                    const float wsm = ws.wsum[a][hh];
                    zb[hh * 32 + s_l] = Qt[0];
#pragma unroll
                    for (int c = 0; c < 3; ++c) zb[64 + c * 64 + hh * 32 + s_l] = Pt[0][c] + (wsm * pi3[c] + ws.z3buf[0][hh][c * 32 + s_l]);
                } else {
                    const int sel = g >> 1, a = 2 * t + sel;
                    float* zb = zrow[sel];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float wsm = ws.wsum[a][h];
                        zb[h * 32 + s_l] = Qt[h];
#pragma unroll
                        for (int c = 0; c < 3; ++c) zb[64 + c * 64 + h * 32 + s_l] = Pt[h][c] + (wsm * pi3[c] + ws.z3buf[sel][h][c * 32 + s_l]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            } else {
            // ---- centre(s) complete: reduce the per-lane partial sums across lane groups, stage in LDS
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    float x = zq[h][k];
                    x += __shfl_xor(x, 16); if (NN >= 16) x += __shfl_xor(x, 32);
                    zq[h][k] = x;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        float y = zp1[h][c][k];
                        y += __shfl_xor(y, 16); if (NN >= 16) y += __shfl_xor(y, 32);
                        zp1[h][c][k] = y;
                    }
                }
            {   // p_j sums: fold the two edge-parity lane groups, stage [h][96] per centre
                const int esub2 = lane / 24, quad2 = lane - 24 * esub2;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        z3a[h][j] += __shfl_down(z3a[h][j], 24);
                        if (NN == 8) z3b[h][j] += __shfl_down(z3b[h][j], 24);
                    }
                    if (esub2 == 0) {
                        st4(&ws.z3buf[0][h][4 * quad2], z3a[h]);
                        if (NN == 8) st4(&ws.z3buf[1][h][4 * quad2], z3b[h]);
                    }
                    z3a[h] = f32x4{0, 0, 0, 0}; z3b[h] = f32x4{0, 0, 0, 0};
                }
            }
            const int slot0 = (FIN && NN >= 16) ? (SUBS > 1 ? sub : (16 * t) / NN) : 0;     // FIN keeps every centre of the iteration staged
            if (g == 0 || (NN == 8 && g == 2)) {
                float* zb = zrow[(NN == 8 && g == 2) ? 1 : slot0];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        zb[h * 32 + 16 * k + e] = zq[h][k];
#pragma unroll
                        for (int c = 0; c < 3; ++c) zb[64 + c * 64 + h * 32 + 16 * k + e] = zp1[h][c][k];
                    }
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int sel = 0; sel < (NN == 8 ? 2 : 1); ++sel) {
                const int a = NN == 8 ? 2 * t + sel : (16 * t) / NN;
                const int i = c0 + a;
                if (FIN) {   // complete the row in place (every lane touches only its own elements)
                    float* zb = zrow[NN == 8 ? sel : slot0];
                    const int c = lane >> 5, s = lane & 31;
                    const float pi0 = pi_pre[sel][0], pi1 = pi_pre[sel][1];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        zb[64 + c * 64 + h * 32 + s] += ws.wsum[a][h] * pi0 + ws.z3buf[sel][h][lane];
                        if (lane < 32) zb[64 + 128 + h * 32 + lane] += ws.wsum[a][h] * pi1 + ws.z3buf[sel][h][64 + lane];
                    }
                } else if (i < N1) {
                    const float* zb = zrow[sel];
                    float* zo = Z + (size_t)i * REC_Z;
                    zo[lane] = zb[lane];
                    const int c = lane >> 5, s = lane & 31;
                    const float pi0 = pi_pre[sel][0];
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        zo[64 + c * 64 + h * 32 + s] = zb[64 + c * 64 + h * 32 + s] + ws.wsum[a][h] * pi0 + ws.z3buf[sel][h][lane];
                    if (lane < 32) {
                        const float pi1 = pi_pre[sel][1];
#pragma unroll
                        for (int h = 0; h < 2; ++h)
                            zo[64 + 128 + h * 32 + lane] = zb[64 + 128 + h * 32 + lane] + ws.wsum[a][h] * pi1 + ws.z3buf[sel][h][64 + lane];
                    }
                }
            }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int k = 0; k < 2; ++k) { zq[h][k] = 0.f; zp1[h][0][k] = zp1[h][1][k] = zp1[h][2][k] = 0.f; }
            __builtin_amdgcn_wave_barrier();
            PHASE_MARK(6);
        }
        }      // !SP
        if (F16 && !FIN) sat = 0.0f;
        }   // !M32
      }   // work item
      }   // sub
      if (FIN && !NODEW) {
#define PESTO_FINR_ON (FIN && !NODEW)
#define PESTO_FINR_W WPB
#define PESTO_FINR_GEN 0
#define PESTO_FINR_ITER fin_iter
#include "pesto_fin_rendezvous.inc"
#undef PESTO_FINR_ON
#undef PESTO_FINR_W
#undef PESTO_FINR_GEN
#undef PESTO_FINR_ITER
        ++fin_iter;
      }
      if (NODEW) {
        // ---- node-wave mode, the item waves' side: the Z rows of this iteration's centres are staged in generation (iteration & 1) of the
        // wave's staging rows; the wave counts itself in XF_READY[generation] and goes straight on to its next items - no rendezvous, no
        // weight fragments, no record stores (whose acknowledgements the next gathers of the same wave would have to wait for). The node
        // waves run a loop of their own in front of this one (round 5: one register allocation per loop).
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);
        const int gen = fin_iter & 1;
        // (twelve item waves: the waves of the second, half-filled tile count in a counter of their own - the node waves start on the
        // first tile as soon as ITS eight waves have staged)
        lds_signal(&sm.xflag[((NWT > 1 && wave_u >= 8) ? XF_READY2 : XF_READY) + gen], lane == 0);
        ++fin_iter;
      }
    }
    if constexpr (TAILR) {
      if (fin_iter > 0) {      // the item waves (the node waves have returned from their loop): finish + prepare of the last iteration's 16 centres
        int tid_t = threadIdx.x;
        asm volatile("" : "+v"(tid_t));
        const int lane = tid_t & 63, wave = tid_t >> 6;
        const bool tail = w_end - last_it < nbx * NE * SUBS;
        const int sstride = tail ? nbx * NE : NE;
        const int base = last_it + jb * NE * (tail ? 1 : SUBS);
        (void)sstride;
        // (the first barrier of the phase also waits for the node waves to END: a wave that has not terminated counts for s_barrier, and
        // they are one tile behind - their last posts precede this phase's, whose counter target continues theirs)
#define PESTO_FINR_ON TAILR
#define PESTO_FINR_W NE
#define PESTO_FINR_GEN ((fin_iter - 1) & 1)
#define PESTO_FINR_ITER (fin_iter - 1)
#include "pesto_fin_rendezvous.inc"
#undef PESTO_FINR_ON
#undef PESTO_FINR_W
#undef PESTO_FINR_GEN
#undef PESTO_FINR_ITER
      }
    }
    if (F16) sat_flush(sat + sat_b, flags);      // (nothing is left here: every probe has been flushed with its centre)
    PHASE_FLUSH();
    TRACE32(51);
}

#ifdef PESTO_DEBUG32
extern "C" int pesto_debug_dump32(float* out, int centre) {
    if (centre >= 0) return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg32_centre), &centre, sizeof(int));
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg32), sizeof(float) * 4096);
}
#endif
void debug_print_phase_cycles() {
#ifdef PESTO_TRACE32
    static unsigned long long tr[4096];
    if (hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_trace32), sizeof tr) == hipSuccess) {
        for (int b = 0; b < 4; ++b) {
            const unsigned long long* t = tr + 1024 * b;
            if (!t[0]) continue;
            fprintf(stderr, "[pesto trace32] nn %d id:delta_cycles of block 0 wave 0 (last launch):", 8 << b);
            for (int k = 1; k < 400 && t[k]; ++k) fprintf(stderr, " %d:%lld", (int)(t[k] >> 56), (long long)((t[k] & 0xffffffffffffffull) - (t[k - 1] & 0xffffffffffffffull)));
            fprintf(stderr, "\n");
        }
    }
#endif
#if defined(PESTO_PROFILE_PHASES) && !defined(PESTO_TRACE32)
    unsigned long long h[12];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase_cycles), sizeof h) == hipSuccess) {
        const char* names[12] = {"setup", "pass1(keys)", "softmax", "p2:L1+L2", "p2:L3(values)", "p2:accumulate", "p2:finalize", "fin:issue", "fin:barrier1",
                                 "fin:rows+barrier2", "fin:compute", "fin:prepare"};
        double tot = 0;
        for (int k = 0; k < 12; ++k) tot += (double)h[k];
        for (int k = 0; k < 12; ++k) fprintf(stderr, "[pesto phase] %-16s %6.2f %%  (%llu)\n", names[k], 100.0 * h[k] / tot, h[k]);
    }
#endif
}

// =============================================================================================== launchers
static int node16_blocks(int N1) {
    const int tiles = (N1 + 15) / 16, pair_chunk = ((tiles + 1) / 2 + 7) / 8;
    return (pair_chunk < 32 ? pair_chunk : 32) * 8;
}
int unpack2_merge_blocks(int n_atoms, int N1) {
    // four slots per thread; the merged launch must stay one wave of workgroups (k_node16 holds 118 KB of LDS: one workgroup per CU)
    const int64_t slots = (int64_t)(n_atoms + 1) * KMAX;
    const int extra = (int)((slots + NODE_WAVES * 64 * 4 - 1) / (NODE_WAVES * 64 * 4));
    return node16_blocks(N1) + extra <= 256 ? extra : 0;
}
void launch_node(hipStream_t st, const float* W, const LayerW* finish, const LayerW* prep, int N1, float* q_state, float* p_state,
                 const float* Z, float* rec_nb, float* rec_cen, int variant, int* flags, Unpack2Args u2) {
    const int tiles = (N1 + 15) / 16, chunk = (tiles + 7) / 8;
    const LayerW dummy{};
    const LayerW& wf = finish ? *finish : dummy;
    const LayerW& wp = prep ? *prep : dummy;
    if (variant == 1) {
        hipLaunchKernelGGL(k_node, dim3((chunk + 3) / 4 * 8), dim3(256), 0, st, W, wf, wp, finish ? 1 : 0, prep ? 1 : 0, N1, q_state, p_state, Z,
                           rec_nb, rec_cen, (const int*)flags);
        return;
    }
    // eight waves = two tiles per iteration; persistent workgroups, at most one per CU (32 per XCD)
    const int n_node = node16_blocks(N1), extra = u2.geo ? unpack2_merge_blocks(u2.n, N1) : 0;
    const dim3 grid(n_node + extra), block(NODE_WAVES * 64);
    hipLaunchKernelGGL(k_node16, grid, block, 0, st, W, wf, wp, finish ? 1 : 0, prep ? 1 : 0, N1, q_state, p_state, Z, rec_nb, rec_cen, flags, u2, n_node);
}

struct EdgeIO {     // per-launch pointers of the edge kernel
    const int* ids_s; const float4* geo; const float* rec_nb; const float* rec_cen; const float* p_state; float* Z; int* flags;
    const float* q_state; float* q_out; float* p_out;      // FIN only: old q state, the other half of the ping-pong pair
    PrepW prep; float* rec_nb_out; float* rec_cen_out;     // FIN only: the next layer's tables and record buffers (null: no prepare phase)
};

template <int NN, int WPB, bool PF, bool F16, bool HY, int TI, bool FIN = false, int NE = WPB, bool M32 = false>
static void launch_edge_k(hipStream_t st, const float* W, const LayerW& lw, int N1, const EdgeIO& io, int max_blocks) {
    constexpr int A = 16 * TI / NN;
    const int n_work = (N1 + A - 1) / A;
    int blocks = ((n_work + 7) / 8 + NE - 1) / NE * 8;     // per-XCD share of the work items, NE (item-processing waves) per workgroup, x 8 XCDs
    if (blocks > max_blocks) blocks = max_blocks / 8 * 8;
    if (blocks < 8) blocks = 8;
    hipLaunchKernelGGL((k_edge<NN, WPB, PF, F16, HY, TI, FIN, NE, M32>), dim3(blocks), dim3(WPB * 64), 0, st, W, lw, N1, n_work, io.ids_s, io.geo, io.rec_nb,
                       io.rec_cen, io.p_state, io.Z, io.flags, io.q_state, io.q_out, io.p_out, io.prep, io.rec_nb_out, io.rec_cen_out);
}

// FINE = false: 64-row work items for every nn; FINE = true: the finest work item that still holds whole centres
template <int WPB, bool PF, bool F16, bool HY = false, bool FINE = false>
static void launch_edge_t(hipStream_t st, const float* W, const LayerW& lw, int N1, const EdgeIO& io, int max_blocks) {
    switch (lw.nn) {
        case 8: launch_edge_k<8, WPB, PF, F16, HY, FINE ? 1 : 4>(st, W, lw, N1, io, max_blocks); break;
        case 16: launch_edge_k<16, WPB, PF, F16, HY, FINE ? 1 : 4>(st, W, lw, N1, io, max_blocks); break;
        case 32: launch_edge_k<32, WPB, PF, F16, HY, FINE ? 2 : 4>(st, W, lw, N1, io, max_blocks); break;
        default: launch_edge_k<64, WPB, PF, F16, HY, 4>(st, W, lw, N1, io, max_blocks); break;
    }
}

// finish-in-edge instantiations of the shipped (hybrid f16-split) kernel: a work item holds at most two centres
template <int WPB, bool FINE>
static void launch_edge_fin(hipStream_t st, const float* W, const LayerW& lw, int N1, const EdgeIO& io, int max_blocks) {
    switch (lw.nn) {
        case 8: launch_edge_k<8, WPB, false, true, true, 1, true>(st, W, lw, N1, io, max_blocks); break;
        case 16: launch_edge_k<16, WPB, false, true, true, FINE ? 1 : 2, true>(st, W, lw, N1, io, max_blocks); break;
        case 32: launch_edge_k<32, WPB, false, true, true, FINE ? 2 : 4, true>(st, W, lw, N1, io, max_blocks); break;
        default: launch_edge_k<64, WPB, false, true, true, 4, true>(st, W, lw, N1, io, max_blocks); break;
    }
}

// Full launches run twelve waves per workgroup in one of two modes:
//   rendezvous mode - all twelve waves process work items, the finish / prepare phase runs behind workgroup rendezvous;
//   node-wave mode  - eight waves process work items, four only finish / prepare (no rendezvous).
// Per item the node-wave mode costs 0.90 / 0.965 / 1.00 / 1.023 of the rendezvous mode at nn = 8 / 16 / 32 / 64 (same-box A/B at 8 x
// 3,000 atoms, where both modes fill their last round equally: few edges per centre = the phase is a large share of the layer and
// pays for its own waves; many edges = the four waves are worth more on the edges). The modes also differ in the ROUND they work in
// (256 workgroups x 12 or 8 waves x items per iteration): the launch takes the mode with the smaller (rounds paid x round size x
// cost per item) - e.g. one structure of 20,000 atoms pays 4 rounds of 3,072 items at nn = 32 in rendezvous mode, 5 of 2,048 in
// node-wave mode (171 -> 141 us). Both modes run the same arithmetic in the same order: results do not depend on the choice.
static double rounds_paid(int n_work, int waves, int subs) {
    const int chunk = (n_work + 7) / 8;                                   // per-XCD share
    const int nbx = (chunk + waves - 1) / waves < 32 ? (chunk + waves - 1) / waves : 32;
    const double round = (double)nbx * waves * subs;                      // items per iteration and XCD
    const double its = chunk / round;
    const double full = (double)(long long)its, rest = its - full;
    // the last, partly filled iteration spreads its items over all workgroups: with two items per wave and iteration it costs half
    const double tail = rest <= 0.0 ? 0.0 : (subs == 2 && rest <= 0.5) ? 0.5 : 1.0;
    return (full + tail) * round;
}
// NW16 (round 5, measured and NOT shipped; profiles/r05_nw16_ab.txt): SIXTEEN-wave node-wave workgroups for the nn = 8 layer at 128
// registers (four waves per SIMD; the kernel fits: 1 - 8 registers spilled), in two forms -
//   -DPESTO_NW16_ITEMS12: twelve item waves + four node waves (three item waves per SIMD; 24 centres per iteration = one full and one half
//       tile for the node waves; 16-row per-wave scratch for the LDS). Item waves alone (node work ablated) 49.9 -> 46.6 us per launch: a
//       third item wave per SIMD buys 7 %, not the 1 / 3 a latency-bound chain would give - and with the node work the launch takes 78 us
//       against 59: the four node waves cannot finish two tiles per iteration;
//   -DPESTO_NW16: eight item waves + TWO TEAMS of four node waves that take the iterations in turn: 59.9 -> 58.7 us per launch; the node
//       work costs 5 us instead of 9, the four extra (polling) waves cost the item waves 4 (item waves alone 49.9 -> 53.8).
// Both give the same bits as the twelve-wave form. What the nn = 8 layer pays for the node work (9 of 59 us) is contention for issue
// slots and the L1, neither the depth of the item chain nor the length of the node chain: DESIGN 4.1h.
static bool node_wave_mode(int nn, int n_work) {
#ifdef PESTO_NW16_ITEMS12
    if (nn == 8) return true;                                      // same round as the rendezvous mode (256 x 12 x 2 centres)
#endif
    const int subs = nn == 64 ? 2 : 1;                                     // items per wave and iteration (two staged centres per wave)
    const double cost = nn == 8 ? 0.87 : nn == 16 ? 0.94 : nn == 32 ? 0.95 : 1.023;      // (re-measured in round 5: nn = 8 56.1 vs 66.9, nn = 16 86.1 vs 91.6, nn = 32 with one pass per tile 144.4 vs 152.2 us)
    return rounds_paid(n_work, 8, subs) * cost < rounds_paid(n_work, 12, subs);
}
// M32: the 32-edge-tile kernel (v_mfma_f32_32x32x16_f16, eight-wave rendezvous workgroups); its gathers use 32-bit buffer offsets
constexpr int M32_MAX_ATOMS = 2000000;      // N1 * REC_CEN * 4 bytes must stay below 2^32
static bool launch_edge_m32(hipStream_t st, const float* W, const LayerW& lw, int N1, const EdgeIO& io, int max_blocks) {
    if (N1 > M32_MAX_ATOMS) return false;
    switch (lw.nn) {
        case 16: launch_edge_k<16, 8, false, true, true, 2, true, 8, true>(st, W, lw, N1, io, max_blocks); return true;
        case 32: launch_edge_k<32, 8, false, true, true, 4, true, 8, true>(st, W, lw, N1, io, max_blocks); return true;
        case 64: launch_edge_k<64, 8, false, true, true, 4, true, 8, true>(st, W, lw, N1, io, max_blocks); return true;
        default: return false;
    }
}
static bool launch_edge_m32_unfused(hipStream_t st, const float* W, const LayerW& lw, int N1, const EdgeIO& io, int max_blocks) {
    if (N1 > M32_MAX_ATOMS) return false;
    switch (lw.nn) {
        case 16: launch_edge_k<16, 8, false, true, true, 2, false, 8, true>(st, W, lw, N1, io, max_blocks); return true;
        case 32: launch_edge_k<32, 8, false, true, true, 4, false, 8, true>(st, W, lw, N1, io, max_blocks); return true;
        case 64: launch_edge_k<64, 8, false, true, true, 4, false, 8, true>(st, W, lw, N1, io, max_blocks); return true;
        default: return false;
    }
}
static void launch_edge_full(hipStream_t st, const float* W, const LayerW& lw, int N1, const EdgeIO& io, int max_blocks, int mode) {
    if (mode == 3 && launch_edge_m32(st, W, lw, N1, io, max_blocks)) return;
    const int a = lw.nn == 64 ? 1 : 2;                                     // centres per wave and work step (nn = 8: one-tile items of two centres; nn = 16 node-wave mode: two one-centre items)
    const bool nw = mode == 0 ? node_wave_mode(lw.nn, (N1 + a - 1) / a) : mode == 2;
    switch (lw.nn) {
#if defined(PESTO_NW16_ITEMS12)
        case 8: if (nw) launch_edge_k<8, 16, false, true, true, 1, true, 12>(st, W, lw, N1, io, max_blocks);
#elif defined(PESTO_NW16)
        case 8: if (nw) launch_edge_k<8, 16, false, true, true, 1, true, 8>(st, W, lw, N1, io, max_blocks);
#else
        case 8: if (nw) launch_edge_k<8, 12, false, true, true, 1, true, 8>(st, W, lw, N1, io, max_blocks);
#endif
                else launch_edge_k<8, 12, false, true, true, 1, true>(st, W, lw, N1, io, max_blocks);
                break;
        // nn = 16, node-wave mode (round 5): ONE-tile items, two per wave and iteration, instead of one two-tile item - a one-tile item reuses
        // the first pass's p_j . r_hat operand in the second pass (ONEP: no second gather of the six p_j pieces, no second projection, split
        // and lane move), which a two-tile item cannot (no registers to keep two operands). Same bits (the fine-item kernels of small
        // launches are this instantiation); same box 90.3 -> 86.1 us per launch (profiles/r05_nn16_onetile_ab.txt).
        case 16: if (nw) launch_edge_k<16, 12, false, true, true, 1, true, 8>(st, W, lw, N1, io, max_blocks);
                 else launch_edge_k<16, 12, false, true, true, 2, true>(st, W, lw, N1, io, max_blocks);
                 break;
        case 32: if (nw) launch_edge_k<32, 12, false, true, true, 4, true, 8>(st, W, lw, N1, io, max_blocks);
                 else launch_edge_k<32, 12, false, true, true, 4, true>(st, W, lw, N1, io, max_blocks);
                 break;
        default: if (nw) launch_edge_k<64, 12, false, true, true, 4, true, 8>(st, W, lw, N1, io, max_blocks);
                 else launch_edge_k<64, 12, false, true, true, 4, true>(st, W, lw, N1, io, max_blocks);
                 break;
    }
}

// variant 0 (default): hybrid first layer (A_j record + per-edge p_j.r block on MFMA), 12 waves per workgroup (3 per SIMD, one
//            workgroup per CU), f16-split MFMA; with q_out / p_out the finish phase runs inside (new state -> q_out / p_out)
// variant 1: everything on exact fp32 MFMA (4 waves per workgroup, explicit cross-tile prefetch), full 2 KB neighbour records
void launch_edge(hipStream_t st, const float* W, const LayerW& lw, int N1, const int* ids_s, const float4* geo,
                 const float* rec_nb, const float* rec_cen, const float* p_state, float* Z, int max_blocks, int variant, int* flags,
                 const float* q_state, float* q_out, float* p_out, const LayerW* next, float* rec_nb_out, float* rec_cen_out, int mode) {
    PrepW pw{};
    if (next) pw = PrepW{next->h_ua, next->h_gc, next->h_n0, next->n_b1s, next->n_bn0, next->n_bn1, next->n_bn2s};      // (the split path's queries carry the softmax scale)
    const EdgeIO io{ids_s, geo, rec_nb, rec_cen, p_state, Z, flags, q_state, q_out, p_out, pw, next ? rec_nb_out : nullptr, next ? rec_cen_out : nullptr};
    // small launches (one structure, or the nn = 8/16 layers of a small batch) cannot fill 256 twelve-wave workgroups:
    // the same kernel body in smaller workgroups spreads them over more CUs
    const int n_work = (N1 + 64 / lw.nn - 1) / (64 / lw.nn);
    if (variant == 1) {
        launch_edge_t<4, true, false>(st, W, lw, N1, io, max_blocks);
    } else if (q_out == nullptr) {
        // unfused f16-split layer (developer mode 4): Z through memory, k_node16 finishes; 32-edge tiles where they exist
        if (mode == 5 || !launch_edge_m32_unfused(st, W, lw, N1, io, 256)) launch_edge_t<12, false, true, true>(st, W, lw, N1, io, 256);
    } else {
        // (the finish phase always runs inside the shipped kernel: q_out / p_out are required)
        // fine work items (one centre each for nn >= 16): when they outnumber the 2,048 wave slots of eight-wave workgroups, twelve
        // waves give (almost) every item its own wave instead of handing half of the waves two
        const int n_fine = lw.nn == 8 ? (N1 + 1) / 2 : N1;
        if (n_work >= 2048 || mode != 0) launch_edge_full(st, W, lw, N1, io, 256, mode);
        else if (n_fine > 2048) launch_edge_fin<12, true>(st, W, lw, N1, io, 256);
        else launch_edge_fin<8, true>(st, W, lw, N1, io, 256);     // 63 KB of constants: one workgroup per CU; fine work items
    }
}

}  // namespace pesto
