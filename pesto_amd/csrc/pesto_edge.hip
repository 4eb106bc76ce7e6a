// pesto_edge.hip - the per-EDGE kernel of the state-update layer on the gfx950 matrix cores: k_edge = one whole layer on the shipped path
// (edges, attention, the layer's output MLPs and the next layer's per-atom records). The item waves' work loop is in this file; the two
// forms of the finish / prepare phase are pesto_fin_rendezvous.inc (rendezvous mode) and pesto_edge_node_waves.inc (node-wave mode).
// Layer overview and the per-atom kernels: pesto_node.hip.
#include "pesto_mfma_common.h"

namespace pesto {

// =============================================================================================== edge kernel
// per-wave LDS scratch of the edge kernel: the 64 edge rows of a work item
struct alignas(16) EdgeWaveScratch {       // 16-byte multiple: the rows are read / written as float4 (ds_read / ds_write_b128)
    int nb[64];            // neighbour id per row
    float geo[5][64];      // r_hat x, y, z, d per row (SoA); row 4 = 1.0 (k = 3 slot of the centre MFMA's B operand)
    float wts[8][64];      // attention weights [h*4 + part][row]: part 0 scalar, 1..3 the vector chunks
    float wsum[8][2];    // per centre: sum over edges of the part-2 weights (multiplies p_i)
    float z3buf[2][2][96];  // [centre sel][h][c*32+s]: sum_e w3[h][e] p_j(e), staged for the final combine
};
// offsets of the NEXT layer's prepare tables (the [U|A], G and nqm fragments / biases of LayerW): the finishing waves of the edge
// kernel write that layer's centre / neighbour records right behind the state update (k_node16's prepare half, same arithmetic)
struct PrepW { int32_t h_ua, h_gc, h_n0, n_b1s, n_bn0, n_bn1, n_bn2, finite; };      // finite: store the centre records finite (the next layer is an nn = 8 layer)
// XCH_FLOATS: tile-state exchange of the prepare phase (rendezvous mode): per 16-centre tile [q0 q1 p00 p01 p10 p11 p20 p21][fg 4][column][4];
// the second tile of a twelve-wave workgroup holds 8 centres (24 per iteration) and is stored compactly: 2048 + 1024 floats
constexpr int XCH_FLOATS = 3072;
// NE = waves that process work items. NE == WPB: every wave does, the finish / prepare phase runs behind workgroup rendezvous.
// NE < WPB ("node waves"): the other WPB - NE waves ONLY finish / prepare, fed through LDS queues without any workgroup barrier:
// two generations of staged Z rows per edge wave and of the 16-centre state exchange.
constexpr int XF_POST = 0, XF_READY = 2, XF_CONSUMED = 4;      // xflag slots: slices posted per tile | rows staged per generation | generations read
template <int WPB, bool HY, bool XCH = false, int NE = WPB>
struct EdgeSmem {
    static constexpr int GEN = (XCH && NE < WPB) ? 2 : 1;
    float w[HY ? EDGE_LDS_FLOATS_HY : EDGE_LDS_FLOATS];
    EdgeWaveScratch ws[NE];
    float zrows[NE][GEN][2][256];   // Zq | Zp staging per centre: two rows per edge wave (and generation)
    float xch[XCH ? (NE < WPB ? 2 * 2048 : XCH_FLOATS) : 4];      // (node waves: two generations)
    // bias of the value network's last layer, four copies per feature: the accumulator tile of feature column e starts as (b, b, b, b) -
    // one ds_read_b128 instead of a 4-byte read + four v_mov per block (16 v_mov per tile). Filled by the kernel's prologue.
    alignas(16) float b3v4[HY ? 256 : 4];
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
    o.a4 = ld4(rp); o.c0 = ld4(rp + 4); o.c1 = ld4(rp + 8); o.c2 = ld4(rp + 12);
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

__device__ __forceinline__ int prod_piece(int lane) { return (lane & 3) ^ ((lane >> 5) << 1); }      // the 16-byte piece of its edge a PRODUCER lane loads (see to_mfma_lanes)
// buffer resources of the arrays the item waves gather from (pesto_mfma_common.h: 32-bit per-lane offsets, uniform parts in SGPRs)
struct EdgeBufs { __amdgpu_buffer_rsrc_t nb, cen, p; };
// per-tile addressing: centre record(s), neighbour record of this lane's edge, geometry. The exact kernels use the pointers; the shipped
// (hybrid) kernels byte offsets: sA / sB = the tile's centre record(s) (wave-uniform: c0 is), vp = this PRODUCER lane's 16 bytes of its
// edge's A_j record (lane = 4 edge + piece, see prod_piece)
struct TileCtx { const float *cenA, *cenB, *recj; int sA, sB, vp; float rx, ry, rz, d, bgA, bgB; };

template <int NN, bool HY = false>
__device__ __forceinline__ TileCtx tile_ctx(int t, int e, int g, int c0, int N1, const EdgeWaveScratch& ws,
                                            const float* __restrict__ rec_nb, const float* __restrict__ rec_cen) {
    TileCtx c;
    const int row = 16 * t + e;
    const int aA = NN == 8 ? 2 * t : (16 * t) / NN;
    if (HY) {
        c.sA = ABL_CEN(min(c0 + aA, N1 - 1)) * (REC_CEN * 4);
        c.sB = ABL_CEN(min(c0 + aA + 1, N1 - 1)) * (REC_CEN * 4);
        c.vp = ABL_NB(ws.nb[16 * t + ((16 * g + e) >> 2)]) * (REC_A * 4) + 16 * prod_piece(16 * g + e);
        c.cenA = c.cenB = c.recj = nullptr;
    } else {
        c.cenA = rec_cen + (size_t)ABL_CEN(min(c0 + aA, N1 - 1)) * REC_CEN;
        c.cenB = rec_cen + (size_t)ABL_CEN(min(c0 + aA + 1, N1 - 1)) * REC_CEN;
        c.recj = rec_nb + (size_t)ABL_NB(ws.nb[row]) * REC_NB;
        c.sA = c.sB = c.vp = 0;
    }
    c.rx = ws.geo[0][row]; c.ry = ws.geo[1][row]; c.rz = ws.geo[2][row]; c.d = ws.geo[3][row];
    const float bg = ws.geo[g == 3 ? 4 : g][row];      // B operand of the centre MFMA: (r_x, r_y, r_z, 1)[k = g], one LDS read
    c.bgA = (NN == 8 && e >= 8) ? 0.0f : bg;
    c.bgB = (NN == 8 && e >= 8) ? bg : 0.0f;
    return c;
}

// ---- hybrid first layer: neighbour terms = A_j (gathered, 512 B) + W[:,161:193] (p_j . r) on the matrix cores
// B operand of the W1P MFMAs for one tile: lane (edge e, kg = g) holds p_j(e) . r_hat for s = 8g .. 8g+7, as f16 hi/lo
// Gathers are issued in a PRODUCER lane layout, lane = 4 * edge + chunk: the four lanes of an edge read 64 contiguous bytes, so
// a quarter-wave (what the vector L1 processes per pass) touches 4 cache lines instead of 16 - the L1's line-request rate,
// not bytes or VALU, bounded this kernel. The MFMA operand layout wants lane = 16 * chunk + edge; values move there with
// ds_bpermute (LDS crossbar, no LDS storage): MFMA lane (e, g) pulls from producer lane 4e + g.
__device__ __forceinline__ float bperm(int src_byte, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_byte, __builtin_bit_cast(int, v)));
}
// Producer lanes hold the 16-byte pieces of an edge in swizzled order - lane 4 e + s holds piece s ^ (e >= 8 ? 2 : 0) - so that the 32
// consumer lanes ds_bpermute serves together pull from 32 different banks (lanes l and l + 32 share one; unswizzled, edges e and e + 8
// collided: 10 of the kernel's 14.8 % SQ_LDS_BANK_CONFLICT, profiles/r03_lds_conflict_ablation.txt). Pure data movement: same bits.
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

template <int NN>
__device__ __forceinline__ L1Raw l1_issue(int fb0, int t, int lane, const TileCtx& tc, const EdgeWaveScratch& ws, const EdgeBufs& B) {
    L1Raw r;
    const int rp = 16 * t + (lane >> 2);               // producer lane: edge rp, piece prod_piece(lane) (32 bytes of p_j, 16 of A_j)
    const int pj = ABL_NB(ws.nb[rp]) * 384 + 32 * prod_piece(lane);      // byte offset of this lane's piece of p_j[x]; y, z: + 128, + 256
    r.x0 = bufld4(B.p, pj); r.x1 = bufld4(B.p, pj + 16); r.y0 = bufld4(B.p, pj + 128); r.y1 = bufld4(B.p, pj + 144);
    r.z0 = bufld4(B.p, pj + 256); r.z1 = bufld4(B.p, pj + 272);
    const int l4 = lane * 4;
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) {
        r.a4[fb] = bufld4(B.nb, tc.vp + (fb0 + fb) * 64);
        r.cA[fb] = bufld1(B.cen, l4 + (fb0 + fb) * 256, tc.sA);
        r.cB[fb] = NN == 8 ? bufld1(B.cen, l4 + (fb0 + fb) * 256, tc.sB) : 0.0f;
    }
    return r;
}

template <int NN>
__device__ __forceinline__ L1Head l1_head(const L1Raw& r, int t, int lane, const TileCtx& tc, const EdgeWaveScratch& ws) {
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
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        hp[j] = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)hp[j]);
        lp[j] = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)lp[j]);
    }
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
__device__ __forceinline__ L1RawAC l1_issue_ac(int fb0, int lane, const TileCtx& tc, const EdgeBufs& B) {
    L1RawAC r;
    const int l4 = lane * 4;
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) {
        r.a4[fb] = bufld4(B.nb, tc.vp + (fb0 + fb) * 64);
        r.cA[fb] = bufld1(B.cen, l4 + (fb0 + fb) * 256, tc.sA);
        r.cB[fb] = NN == 8 ? bufld1(B.cen, l4 + (fb0 + fb) * 256, tc.sB) : 0.0f;
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
#ifdef PESTO_DEV_TIMELINE      // developer build (profiles/dev/timeline.py): wall-clock stamps (100 MHz) of every workgroup of the first 64 layer launches
__device__ unsigned long long g_tl[64][256][16];
__device__ int g_tl_launch;
#define PESTO_TL(k) { if ((threadIdx.x & 63) == 0 && tl_l < 64 && blockIdx.x < 256) g_tl[tl_l][blockIdx.x][k] = wall_clock64(); }
#else
#define PESTO_TL(k) {}
#endif
// Two families of instantiations: F16 = true, the shipped f16-split kernels (hybrid first layer HY, finish / prepare phase inside FIN,
// 8- or 12-wave workgroups); F16 = false, the exact fp32 kernels (4-wave workgroups, explicit cross-tile prefetch PF, Z through memory).
template <int NN, int WPB, bool F16, int TI = 4, int NE = WPB>
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
    constexpr bool PF = !F16, HY = F16, FIN = F16;
    constexpr int A = 16 * TI / NN;            // whole centres per work item
    constexpr int TPC = NN >= 16 ? NN / 16 : 1;   // tiles per centre
    static_assert(TI >= TPC && TI % TPC == 0 && A >= 1 && (!PF || TI == 4), "a work item holds whole centres");
    static_assert(!FIN || (HY && A <= 2 && WPB >= 8), "finish phase: at most two staged centres per wave, four waves per 16-centre tile");
    // FIN: work items a wave processes between two finish phases - as many as its two staging rows hold centres (nn = 64: two
    // one-centre items), which halves the number of workgroup rendezvous
    constexpr int SUBS = FIN ? 2 / A : 1;
    constexpr bool NODEW = FIN && NE < WPB;      // node-wave mode: waves NE.. finish / prepare only
    static_assert(NE == WPB || (FIN && WPB - NE == 4 && NE * A * SUBS == 16), "node waves: four of them, one 16-centre tile per iteration");
    __shared__ EdgeSmem<WPB, HY, FIN, NE> sm;
    if (threadIdx.x < 8) sm.xflag[threadIdx.x] = 0;
#ifdef PESTO_DEV_TIMELINE
    const int tl_l = F16 ? *(volatile int*)&g_tl_launch : 64;
    if (threadIdx.x == 0) PESTO_TL(0)
#endif
    // The edge rows (neighbour id, geometry) of the wave's FIRST work item are requested before the layer constants are staged, so that
    // their round trip runs under that copy instead of behind the workgroup barrier - a small launch (one structure) is one or two
    // items per wave deep and pays every such latency in full (one-structure forward: see DESIGN 4.1f).
    int first_nb = 0;
    float4 first_geo = float4{0.f, 0.f, 0.f, 0.f};
    bool first_valid = false, first_item = false;
    {
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
            first_geo = geo[src0];      // (plain loads: once per launch)
        }
    }
    {   // layer constants -> LDS (once per workgroup; workgroups are persistent over work items)
        const f32x4* src = reinterpret_cast<const f32x4*>(W + (F16 ? lw.e_lds16 : lw.e_lds));
        copy_to_lds<(HY ? EDGE_LDS_FLOATS_HY : EDGE_LDS_FLOATS) / 4, WPB * 64>(reinterpret_cast<f32x4*>(sm.w), src, (int)threadIdx.x);
    }
    if constexpr (HY) {
        if (threadIdx.x < 64) {
            const float b = W[lw.e_lds16 + EL_B3V + threadIdx.x];
            st4(&sm.b3v4[4 * threadIdx.x], f32x4{b, b, b, b});
        }
    }
    if (first_item) {     // rows of the first item -> the wave's scratch (no register survives into the work loop)
        auto& ws0 = sm.ws[threadIdx.x >> 6];
        const int l0 = threadIdx.x & 63;
        ws0.nb[l0] = first_valid ? first_nb : 0;
        ws0.geo[0][l0] = first_valid ? first_geo.x : 0.f; ws0.geo[1][l0] = first_valid ? first_geo.y : 0.f;
        ws0.geo[2][l0] = first_valid ? first_geo.z : 0.f; ws0.geo[3][l0] = first_valid ? first_geo.w : 0.f; ws0.geo[4][l0] = 1.0f;
    }
    __syncthreads();
#ifdef PESTO_DEV_TIMELINE
    if (threadIdx.x == 0) PESTO_TL(1)
#endif
    const EdgeBufs eb{make_rsrc(rec_nb), make_rsrc(rec_cen), make_rsrc(p_state)};
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
    // range guard of the f16-split path (sat_probe), flushed per centre (sat_flush_at). sat_b: the second centre of a two-centre item
    // whose centres are different tiles (nn = 16 / 32); for nn = 8 the two centres of a tile are different lanes
    float sat = 0.0f, sat_b = 0.0f;
    constexpr bool SAT2 = F16 && A == 2 && NN >= 16;
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
#include "pesto_edge_node_waves.inc"
      }
    }
    for (int it_start = xcd * chunk; it_start < w_end; it_start += nbx * NE * SUBS) {
      // the lane-derived values of the work loop (indices, LDS addresses, masks) are re-derived per iteration from an opaque copy of the
      // thread index: as loop invariants they would stay in registers across the finish / prepare phase below, which needs them for
      // weight fragments (forty registers; the fragments of that phase otherwise spill in front of its rendezvous)
      int tid_i = threadIdx.x;
      if (FIN) asm volatile("" : "+v"(tid_i));
      const int lane = tid_i & 63, wave = tid_i >> 6;
      const int e = lane & 15, g = lane >> 4;
      // wave priority by age class (waves 0-3 / 4-7 / 8-11 of a workgroup = oldest / middle / youngest wave of their SIMD). At equal
      // priority the arbiter prefers the oldest wave: in a full nn = 64 launch wave 0 reaches the rendezvous of its twelve-wave workgroup
      // 20 - 28 thousand cycles (~25 % of an iteration) ahead of the slowest wave and its SIMD runs on two waves for that long. The
      // priority of a wave's MFMA bursts therefore grows with its youth (1 / 2 / 3; outside the bursts 0 for everybody, as before):
      // nn = 64 -1.2 ... -2.5 % per launch on three boxes (profiles/r05_prio_age_ab.txt); the reverse order (3, 2, 1), a raised base level
      // for the young waves and the same table on the one-tile items' two-pass path (nn = 8 / 16) gain nothing. Scheduling only: same bits.
      const int wave_p = __builtin_amdgcn_readfirstlane(wave) >> 2;
      (void)wave_p;
#define PESTO_PRIO_HI() prio_by_age<1, 2, 3>(wave_p)       // one pass per tile (nn = 32 / 64)
#define PESTO_PRIO_LO() __builtin_amdgcn_s_setprio(0)
#define PESTO_PRIO_HI2() __builtin_amdgcn_s_setprio(1)      // the two-pass path: one-tile items of nn = 8 / 16, the exact kernels
#define PESTO_PRIO_LO2() __builtin_amdgcn_s_setprio(0)
      const int wslot = (NODEW && wave >= NE) ? 0 : wave;      // (node waves never touch the per-wave scratch)
      auto& ws = sm.ws[wslot];
      float (*zrow)[256] = sm.zrows[wslot][NODEW ? (fin_iter & 1) : 0];
      const bool tail = w_end - it_start < nbx * NE * SUBS;
      const int sstride = tail ? nbx * NE : NE;
      const int base = it_start + jb * NE * (tail ? 1 : SUBS);
      // node-wave mode: this generation of staging rows was last used two iterations ago - the node waves must have read it
      if (NODEW && wave < NE && fin_iter >= 2) {
          lds_wait_ge(&sm.xflag[XF_CONSUMED], 4 * (fin_iter - 1));
      }
#pragma unroll 1
      for (int sub = 0; sub < SUBS; ++sub) {
      const int work = base + sub * sstride + wave;
      bool item_on = work < w_end && (!NODEW || wave < NE);
      if constexpr (!F16 && !FIN) {      // the exact kernels as AUTO's fp32 repeat: items without a centre of a flagged structure are skipped
          if (item_on && only_fl) item_on = rows_flagged(flags, work * A, A, N1, lane);
      }
      if (item_on) {
        // (wave-uniform, and scalar for the compiler on the shipped path: the centre-record offsets below are then SGPR arithmetic)
        const int c0 = F16 ? __builtin_amdgcn_readfirstlane(work * A) : work * A;
        if (!(it_start == xcd * chunk && sub == 0)) {   // rows of this work item: lane = row (the first item's rows are staged already)
            const int a = lane / NN, c = lane % NN, i = c0 + a;
            const bool valid = i < N1 && lane < 16 * TI;
            const unsigned src = (unsigned)min(i, N1 - 1) * KMAX + c;       // unconditional loads, select afterwards
            const int nbv = __builtin_bit_cast(int, bufld1(make_rsrc(ids_s), (int)(src * 4u)));
            const f32x4 gg4 = bufld4(make_rsrc(geo), (int)(src * 16u));
            const float4 gg = float4{gg4[0], gg4[1], gg4[2], gg4[3]};
            ws.nb[lane] = valid ? nbv : 0;
            ws.geo[0][lane] = valid ? gg.x : 0.f; ws.geo[1][lane] = valid ? gg.y : 0.f; ws.geo[2][lane] = valid ? gg.z : 0.f;
            ws.geo[3][lane] = valid ? gg.w : 0.f; ws.geo[4][lane] = 1.0f;
        }
        __builtin_amdgcn_wave_barrier();

        // ------------------------------------------------------------------ pass 1: keys -> logits (eqkm, epkm)
        constexpr bool ONEP = TI == 1 && !PF && HY;      // one-tile items: see l1_issue_ac
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
        // structures/s (+3.6 %). nn = 16 tried the same: no gain (85.2 vs 84.8 us), it stays on the two-pass one-tile path.
        constexpr bool SP = F16 && NN >= 32;
        L1RawAC rac2;
        f16x8 pr_h, pr_l;
        (void)rac2; (void)pr_h; (void)pr_l;
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
                float Qv[6];
                if constexpr (F16) {      // six floats at byte 2048 (+ 24 for the vector queries) of the centre record: the centre as the scalar offset
                    const int aT = NN == 8 ? 2 * t : (16 * t) / NN;
                    const int sq = ABL_CEN(min(c0 + aT, N1 - 1)) * (REC_CEN * 4) + 512 * 4;
                    int vq = g == 0 ? 0 : 24;
                    if (NN == 8) vq += (e >> 3) * ((ABL_CEN(min(c0 + aT + 1, N1 - 1)) - ABL_CEN(min(c0 + aT, N1 - 1))) * (REC_CEN * 4));      // the tile's second centre
                    const f32x4 qa = bufld4(eb.cen, vq, sq);
                    const f32x2 qb = bufld2(eb.cen, vq + 16, sq);
                    Qv[0] = qa[0]; Qv[1] = qa[1]; Qv[2] = qa[2]; Qv[3] = qa[3]; Qv[4] = qb[0]; Qv[5] = qb[1];
                } else {
                    const int aMine = NN == 8 ? 2 * t + (e >> 3) : (16 * t) / NN;
                    const float* Qp = rec_cen + (size_t)ABL_CEN(min(c0 + aMine, N1 - 1)) * REC_CEN + 512 + (g == 0 ? 0 : 6);
#pragma unroll
                    for (int k = 0; k < 6; ++k) Qv[k] = Qp[k];
                }
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
                        for (int c = 0; c < 3; ++c) pi3[c] = bufld1(eb.p, (16 * (g & 1) + e) * 4 + c * 128, ic * 384);
                    }
                    const L1Raw raw = l1_issue<NN>(0, t, lane, tcc, ws, eb);
                    __builtin_amdgcn_sched_barrier(0);
                    L1Head hd = l1_head<NN>(raw, t, lane, tcc, ws);
                    __builtin_amdgcn_sched_barrier(0);
                    const L1RawAC rac = l1_issue_ac<NN>(4, lane, tcc, eb);      // the second half's A_j chunks / centre columns: in flight under the key networks
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
                        for (int i2 = 0; i2 < 4; ++i2) pv[i2] = bufld4(eb.p, nbj[i2] * 384 + 16 * quad);
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
                        for (int i2 = 0; i2 < 4; ++i2) pv[i2] = bufld4(eb.p, nbj[i2] * 384 + 16 * quad);
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
                    L1Raw raw = l1_issue<NN>(0, 0, lane, tcc, ws, eb);
#pragma unroll 1
                    for (int t = 0; t < TI; ++t) {
                        if (SAT2 && t == TI / 2) { const float tmp = sat; sat = sat_b; sat_b = tmp; }      // the second centre's tiles start
                        L1Head hd = l1_head<NN>(raw, t, lane, tcc, ws);
                        __builtin_amdgcn_sched_barrier(0);
                        if (ONEP) { rac2 = l1_issue_ac<NN>(4, lane, tcc, eb); pr_h = hd.fh; pr_l = hd.fl; }
                        if (t < TI - 1) {
                            tcc = tile_ctx<NN, HY>(t + 1, e, g, c0, N1, ws, rec_nb, rec_cen);
                            raw = l1_issue<NN>(0, t + 1, lane, tcc, ws, eb);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        f32x4 h1[4];
                        // wave priority: a wave inside its MFMA burst (first-layer tail, key / value networks) goes ahead of waves that are in
                        // VALU / LDS phases (softmax, weighted sums, finalize) - measured +3.8 % (levels 1..3 alike)
                        PESTO_PRIO_HI2();
                        l1_tail(hd, 0, lane, g, sm.w + EL_W1P, sm.w + EL_WD, h1, sat);
                        keys_of_tile(t, h1);
                        PESTO_PRIO_LO2();
                    }
                    if (SAT2) { const float tmp = sat; sat = sat_b; sat_b = tmp; }      // sat: first centre again, sat_b: second
                }
            }
        }
        if constexpr (!SP) {
        float lg[4][2];
#pragma unroll
        for (int t = 0; t < TI; ++t) { lg[t][0] = ws.wts[g][16 * t + e]; lg[t][1] = ws.wts[4 + g][16 * t + e]; }
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
                    const int aT = NN == 8 ? 2 * t : (16 * t) / NN;
                    const int ic = ABL_CEN(min(c0 + aT, N1 - 1));
                    int vpi = (16 * (g & 1) + e) * 4;
                    if (NN == 8) vpi += (g >> 1) * ((ABL_CEN(min(c0 + aT + 1, N1 - 1)) - ic) * 384);      // lane rows 2, 3: the tile's second centre
#pragma unroll
                    for (int c = 0; c < 3; ++c) pi3[c] = bufld1(eb.p, vpi + c * 128, ic * 384);
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
            {
                int nbj[4];
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) nbj[i2] = ABL_NB(ws.nb[16 * t + 2 * i2 + (esub & 1)]);
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) pv[i2] = bufld4(eb.p, nbj[i2] * 384 + 16 * quad);
            }
            f32x4 h1[4];
            if (PF) {
#pragma unroll
                for (int fbl = 0; fbl < 4; ++fbl)
                    h1[fbl] = l1_compute<NN>(pre[fbl], 4 + fbl, g, tc.bgA, tc.bgB, sm.w + EL_WD, tc.d, tc.rx, tc.ry, tc.rz);
            } else {
                L1Head hd;
                if (ONEP) {
                    hd = l1_head_ac<NN>(rac2, pr_h, pr_l, lane, tc);
                } else {
                    const L1Raw raw = l1_issue<NN>(4, t, lane, tc, ws, eb);
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
            {
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
            __builtin_amdgcn_sched_barrier(0);
            {   // second half of the tile's edges: these loads land during the MFMA phase
                int nbj[4];
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) nbj[i2] = ABL_NB(ws.nb[16 * t + 8 + 2 * i2 + (esub & 1)]);
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) pv[i2] = bufld4(eb.p, nbj[i2] * 384 + 16 * quad);
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
                if constexpr (HY) {
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
            }
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
        }
        }      // !SP
        if (F16 && !FIN) sat = 0.0f;
      }   // work item
      }   // sub
      if (FIN && !NODEW) {
#include "pesto_fin_rendezvous.inc"
        ++fin_iter;
      }
      if (NODEW) {
        // ---- node-wave mode, the item waves' side: the Z rows of this iteration's centres are staged in generation (iteration & 1) of the
        // wave's staging rows; the wave counts itself in XF_READY[generation] and goes straight on to its next items - no rendezvous, no
        // weight fragments, no record stores (whose acknowledgements the next gathers of the same wave would have to wait for). The node
        // waves run a loop of their own in front of this one (round 5: one register allocation per loop).
        lds_signal(&sm.xflag[XF_READY + (fin_iter & 1)], lane == 0);
        ++fin_iter;
      }
    }
    if (F16) sat_flush(sat + sat_b, flags);      // (nothing is left here: every probe has been flushed with its centre)
#ifdef PESTO_DEV_TIMELINE
    if (threadIdx.x == 0) { PESTO_TL(2) if (blockIdx.x == 0 && F16) atomicAdd(&g_tl_launch, 1); }
    if (threadIdx.x == 64 * 7) PESTO_TL(4)
#endif
}

// =============================================================================================== launchers
struct EdgeIO {     // per-launch pointers of the edge kernel
    const int* ids_s; const float4* geo; const float* rec_nb; const float* rec_cen; const float* p_state; float* Z; int* flags;
    const float* q_state; float* q_out; float* p_out;      // FIN only: old q state, the other half of the ping-pong pair
    PrepW prep; float* rec_nb_out; float* rec_cen_out;     // FIN only: the next layer's tables and record buffers (null: no prepare phase)
};

template <int NN, int WPB, bool F16, int TI, int NE = WPB>
static void launch_edge_k(hipStream_t st, const float* W, const LayerW& lw, int N1, const EdgeIO& io, int max_blocks) {
    constexpr int A = 16 * TI / NN;
    const int n_work = (N1 + A - 1) / A;
    int blocks = ((n_work + 7) / 8 + NE - 1) / NE * 8;     // per-XCD share of the work items, NE (item-processing waves) per workgroup, x 8 XCDs
    if (blocks > max_blocks) blocks = max_blocks / 8 * 8;
    if (blocks < 8) blocks = 8;
    hipLaunchKernelGGL((k_edge<NN, WPB, F16, TI, NE>), dim3(blocks), dim3(WPB * 64), 0, st, W, lw, N1, n_work, io.ids_s, io.geo, io.rec_nb,
                       io.rec_cen, io.p_state, io.Z, io.flags, io.q_state, io.q_out, io.p_out, io.prep, io.rec_nb_out, io.rec_cen_out);
}

// the exact fp32 kernels (PESTO_PRECISION_FP32, AUTO's repeat): four-wave workgroups, 64-row work items for every nn
static void launch_edge_exact(hipStream_t st, const float* W, const LayerW& lw, int N1, const EdgeIO& io, int max_blocks) {
    switch (lw.nn) {
        case 8: launch_edge_k<8, 4, false, 4>(st, W, lw, N1, io, max_blocks); break;
        case 16: launch_edge_k<16, 4, false, 4>(st, W, lw, N1, io, max_blocks); break;
        case 32: launch_edge_k<32, 4, false, 4>(st, W, lw, N1, io, max_blocks); break;
        default: launch_edge_k<64, 4, false, 4>(st, W, lw, N1, io, max_blocks); break;
    }
}

// small launches of the shipped kernel (one structure, or the nn = 8 / 16 layers of a small batch): the finest work item that still
// holds whole centres, so that the launch spreads over more waves and CUs
template <int WPB>
static void launch_edge_fine(hipStream_t st, const float* W, const LayerW& lw, int N1, const EdgeIO& io, int max_blocks) {
    switch (lw.nn) {
        case 8: launch_edge_k<8, WPB, true, 1>(st, W, lw, N1, io, max_blocks); break;
        case 16: launch_edge_k<16, WPB, true, 1>(st, W, lw, N1, io, max_blocks); break;
        case 32: launch_edge_k<32, WPB, true, 2>(st, W, lw, N1, io, max_blocks); break;
        default: launch_edge_k<64, WPB, true, 4>(st, W, lw, N1, io, max_blocks); break;
    }
}

// Full launches run twelve waves per workgroup in one of two modes:
//   rendezvous mode - all twelve waves process work items, the finish / prepare phase runs behind workgroup rendezvous;
//   node-wave mode  - eight waves process work items, four only finish / prepare (no rendezvous).
// Per item the node-wave mode costs 0.87 / 0.94 / 0.95 / 1.023 of the rendezvous mode at nn = 8 / 16 / 32 / 64 (same-box A/B at 8 x
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
static bool node_wave_mode(int nn, int n_work) {
    const int subs = nn == 64 ? 2 : 1;                                     // items per wave and iteration (two staged centres per wave)
    const double cost = nn == 8 ? 0.87 : nn == 16 ? 0.94 : nn == 32 ? 0.95 : 1.023;      // (re-measured in round 5: nn = 8 56.1 vs 66.9, nn = 16 86.1 vs 91.6, nn = 32 with one pass per tile 144.4 vs 152.2 us)
    return rounds_paid(n_work, 8, subs) * cost < rounds_paid(n_work, 12, subs);
}
// mode (pesto_debug_edge_mode, test hook): 0 = chosen per launch, 1 = rendezvous mode, 2 = node-wave mode
static void launch_edge_full(hipStream_t st, const float* W, const LayerW& lw, int N1, const EdgeIO& io, int max_blocks, int mode) {
    const int a = lw.nn == 64 ? 1 : 2;                                     // centres per wave and work step (nn = 8: one-tile items of two centres; nn = 16 node-wave mode: two one-centre items)
    const bool nw = mode == 0 ? node_wave_mode(lw.nn, (N1 + a - 1) / a) : mode == 2;
    switch (lw.nn) {
        case 8: if (nw) launch_edge_k<8, 12, true, 1, 8>(st, W, lw, N1, io, max_blocks);
                else launch_edge_k<8, 12, true, 1>(st, W, lw, N1, io, max_blocks);
                break;
        // nn = 16, node-wave mode: ONE-tile items, two per wave and iteration, instead of one two-tile item - a one-tile item reuses the
        // first pass's p_j . r_hat operand in the second pass (ONEP: no second gather of the six p_j pieces, no second projection, split
        // and lane move), which a two-tile item cannot (no registers to keep two operands). Same bits (the fine-item kernels of small
        // launches are this instantiation); same box 90.3 -> 86.1 us per launch (profiles/r05_nn16_onetile_ab.txt).
        case 16: if (nw) launch_edge_k<16, 12, true, 1, 8>(st, W, lw, N1, io, max_blocks);
                 else launch_edge_k<16, 12, true, 2>(st, W, lw, N1, io, max_blocks);
                 break;
        case 32: if (nw) launch_edge_k<32, 12, true, 4, 8>(st, W, lw, N1, io, max_blocks);
                 else launch_edge_k<32, 12, true, 4>(st, W, lw, N1, io, max_blocks);
                 break;
        default: if (nw) launch_edge_k<64, 12, true, 4, 8>(st, W, lw, N1, io, max_blocks);
                 else launch_edge_k<64, 12, true, 4>(st, W, lw, N1, io, max_blocks);
                 break;
    }
}

// variant 0 (default): the shipped f16-split kernel - hybrid first layer (A_j record + per-edge p_j.r block on MFMA), one workgroup per
//            CU; the finish / prepare phase runs inside (new state -> q_out / p_out, the next layer's records -> rec_*_out)
// variant 1: everything on exact fp32 MFMA (4 waves per workgroup, explicit cross-tile prefetch), full 2 KB neighbour records, Z -> memory
void launch_edge(hipStream_t st, const float* W, const LayerW& lw, int N1, const int* ids_s, const float4* geo,
                 const float* rec_nb, const float* rec_cen, const float* p_state, float* Z, int max_blocks, int variant, int* flags,
                 const float* q_state, float* q_out, float* p_out, const LayerW* next, float* rec_nb_out, float* rec_cen_out, int mode) {
    PrepW pw{};
    if (next) pw = PrepW{next->h_ua, next->h_gc, next->h_n0, next->n_b1s, next->n_bn0, next->n_bn1, next->n_bn2s, next->nn == 8 ? 1 : 0};      // (the split path's queries carry the softmax scale)
    const EdgeIO io{ids_s, geo, rec_nb, rec_cen, p_state, Z, flags, q_state, q_out, p_out, pw, next ? rec_nb_out : nullptr, next ? rec_cen_out : nullptr};
    if (variant == 1) { launch_edge_exact(st, W, lw, N1, io, max_blocks); return; }
    if (q_out == nullptr || p_out == nullptr) { fprintf(stderr, "pesto: launch_edge: the shipped kernel needs the output half of the state pair\n"); abort(); }
    // small launches (one structure, or the nn = 8/16 layers of a small batch) cannot fill 256 twelve-wave workgroups:
    // the same kernel body in smaller workgroups spreads them over more CUs
    const int n_work = (N1 + 64 / lw.nn - 1) / (64 / lw.nn);
    // fine work items (one centre each for nn >= 16): when they outnumber the 2,048 wave slots of eight-wave workgroups, twelve
    // waves give (almost) every item its own wave instead of handing half of the waves two
    const int n_fine = lw.nn == 8 ? (N1 + 1) / 2 : N1;
    if (n_work >= 2048 || mode != 0) launch_edge_full(st, W, lw, N1, io, 256, mode);
    else if (n_fine > 2048) launch_edge_fine<12>(st, W, lw, N1, io, 256);
    else launch_edge_fine<8>(st, W, lw, N1, io, 256);     // 63 KB of constants: one workgroup per CU
}

}  // namespace pesto

#ifdef PESTO_DEV_TIMELINE
// developer build only: copies the stamps out (out: 64 x 256 x 8 uint64) and rewinds the launch counter; returns the launches seen
extern "C" int pesto_dev_timeline(unsigned long long* out) {
    int n = 0;
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(&n, HIP_SYMBOL(pesto::g_tl_launch), sizeof(int));
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(pesto::g_tl), sizeof(unsigned long long) * 64 * 256 * 16);
    const int zero = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(pesto::g_tl_launch), &zero, sizeof(int));
    return n;
}
#endif
