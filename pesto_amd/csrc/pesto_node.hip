// pesto_node.hip - the per-ATOM kernels of the state-update layer on the gfx950 matrix cores (k_node: exact fp32; k_node16: f16 split).
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
#include "pesto_mfma_common.h"

namespace pesto {

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
                    if (ob < 8) st4_wt_finite(cen + (ob + j) * 64 + 3 * 16 + 4 * g, a[j]);      // (write-through: 63 MB per launch must not sit dirty in L2 at the kernel boundary)
                    else st4_wt(nb + (ob + j - 8) * 16 + 4 * g, a[j]);                   // A_j[16 fb + 4g + r]
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
                    for (int c = 0; c < 3; ++c) st4_wt_finite(cen + (ob + j) * 64 + c * 16 + 4 * g, a[j][c]);
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
            if (valid) st4_wt(cen + 512 + 4 * g, qq[0]);
        }
        if (valid) sat_flush_at(sat, flags, i);      // (this lane's probes cover MFMA column e = atom i of the tile)
        sat = 0.0f;
    }
}

// =============================================================================================== launcher
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

}  // namespace pesto
