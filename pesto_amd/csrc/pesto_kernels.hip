// pesto_kernels.hip - gfx950 kernels of the PeSTo forward pass: embedding, geometry unpack, the v1
// state-update layer (LDS-tiled fp32 VALU; the MFMA layer lives in pesto_node.hip / pesto_edge.hip), residue pool + decoder.
//
// Math restated from the reference (file:line relative to /root/reference):
//   embedding              model/model.py:34
//   geometry + sink        src/model_operations.py:6-22
//   state-update layer     src/model_operations.py:87-154, :225-242
//   residue pool + decoder src/model_operations.py:197-213, model/model.py:46-50
#include <hip/hip_runtime.h>

#include "pesto_kernels.h"

namespace pesto {

__device__ __forceinline__ float elu(float x) { return x > 0.0f ? x : expf(x) - 1.0f; }

// One output column `s` of y = x Wt + b for a group of 32 lanes; x lives in LDS (read as broadcast),
// Wt[in][out] in global memory (lanes read consecutive floats).
// The k loop is a chain of dependent FMAs fed by two loads per step; left as a scalar loop every step waited for its own LDS / L2
// round trip (the residue-pool kernel: 576 steps = 22 us). Steps are taken U at a time: the inputs as 16-byte LDS reads, the U
// weight loads (L2 hits, ~250 ns a round trip whatever their number) issued together, the FMAs in the ORIGINAL order (same bits).
template <int U>
__device__ __forceinline__ float g32_steps(const float* __restrict__ w, int n_out, const float* x, int& k, int n_in, float acc) {
    for (; k + U <= n_in; k += U) {
        float wv[U];
#pragma unroll
        for (int j = 0; j < U; ++j) wv[j] = w[(k + j) * n_out];
#pragma unroll
        for (int j = 0; j < U; j += 4) {
            const float4 xv = *reinterpret_cast<const float4*>(x + k + j);
            acc += xv.x * wv[j]; acc += xv.y * wv[j + 1]; acc += xv.z * wv[j + 2]; acc += xv.w * wv[j + 3];
        }
    }
    return acc;
}
__device__ __forceinline__ float g32_linear(const float* __restrict__ W, const LinearW l, const float* x, int s) {
    if (s >= l.n_out) return 0.0f;
    float acc = l.b >= 0 ? W[l.b + s] : 0.0f;
    const float* w = W + l.w + s;
    const int n_out = l.n_out;
    int k = 0;
    if ((reinterpret_cast<size_t>(x) & 15) == 0) {
        acc = g32_steps<8>(w, n_out, x, k, l.n_in, acc);      // (32 at a time: 18.3 instead of 19.0 us for one structure's pool kernel, but the
                                                              //  embed / pool-logit kernels of a full batch lose 20 % to the registers)
    }
    for (; k < l.n_in; ++k) acc += x[k] * w[k * n_out];
    return acc;
}

// segment bounds, zero-initialised (a memset shared with other per-forward words instead of an init kernel): lo_enc[r] = max over the
// residue's atoms of (0x7fffffff - i), i.e. 0x7fffffff - (min atom index), hi[r] = max + 1; 0 / 0 = no atom (integer atomics: deterministic)
__device__ __forceinline__ void seg_bound_atom(int i, int R, const int* __restrict__ roa, int* __restrict__ lo_enc, int* __restrict__ hi,
                                               int* __restrict__ err_flag) {
    const int r = roa[i];
    if (r < 0 || r >= R) { atomicOr(err_flag, 2); return; }
    atomicMax(&lo_enc[r], 0x7fffffff - i);
    atomicMax(&hi[r], i + 1);
}
// ------------------------------------------------------------------------------------------------ embedding
// q[i+1][:] = em(q0[i][:]) ; 8 atoms per 256-thread block, 32 lanes per atom.  model/model.py:34
__global__ __launch_bounds__(256) void k_embed(const float* __restrict__ W, MlpW em, int N, int nq, int n0,
                                               const float* __restrict__ q0, float* __restrict__ q_state, float* __restrict__ p_zero,
                                               ClearArgs clr, SatCtx sc) {
    // the forward's per-call words (error flags, max(D) bit patterns, segment bounds of the pool layer) are cleared by workgroup 0
    // of this FIRST launch instead of by fill launches in front of it: nothing in this kernel reads or writes them otherwise
    if (blockIdx.x == 0) {
        for (int k = threadIdx.x; k < clr.n0; k += 256) clr.p0[k] = 0;
        for (int k = threadIdx.x; k < clr.n1; k += 256) clr.p1[k] = 0;
        for (int k = threadIdx.x; k < clr.n2; k += 256) clr.p2[k] = 0;
        if (sc.flags && threadIdx.x == 0) *reinterpret_cast<SatCtx*>(sc.flags + SATCTX_OFFSET_INTS) = sc;      // the range guard's context (pesto_kernels.h)
    }
    __shared__ __attribute__((aligned(16))) float xs[8][512];
    __shared__ __attribute__((aligned(16))) float hs[8][64];
    const int g = threadIdx.x >> 5, s = threadIdx.x & 31;
    const int i = blockIdx.x * 8 + g;
    const int ic = (i < N ? i : N - 1) % nq;   // nq < N: frames of a trajectory share one q0 (row i of every frame)
    for (int k = s; k < n0; k += 32) xs[g][k] = q0[(size_t)ic * n0 + k];
    __syncthreads();
    float v = g32_linear(W, em.l[0], xs[g], s);
    if (em.depth == 3) {
        hs[g][s] = elu(v);
        __syncthreads();
        v = g32_linear(W, em.l[1], hs[g], s);
        hs[g][32 + s] = elu(v);
        __syncthreads();
        v = g32_linear(W, em.l[2], hs[g] + 32, s);
    }
    if (i < N) q_state[(size_t)(i + 1) * S + s] = v;
    // p_zero (optional): p0 = zeros (model/model.py:37) and the sink rows of q and p (model_operations.py:17) written here instead of
    // by two fill launches in front of the forward
    if (p_zero) {
        if (i < N) {
#pragma unroll
            for (int c = 0; c < 3; ++c) p_zero[(size_t)(i + 1) * 96 + 32 * c + s] = 0.0f;
        }
        if (blockIdx.x == 0 && g == 0) {
            q_state[s] = 0.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c) p_zero[32 * c + s] = 0.0f;
        }
    }
}

// conditioning trigger of PESTO_PRECISION_AUTO for zero-padded neighbour slots (SatCtx::pad_trigger): called by ONE lane of a wave that met
// a padded slot; row = state row (1-based) of an atom of the structure. The context was stored behind the flags word by the embed launch.
__device__ __forceinline__ void pad_flag_at(int* __restrict__ err_flag, int64_t row) {
    const SatCtx sc = *reinterpret_cast<const SatCtx*>(err_flag + SATCTX_OFFSET_INTS);
    if (!sc.pad_trigger || !sc.sflags || row < 1) return;
    atomicOr(err_flag, 4);
    atomicOr(sc.sflags + (sc.seg_of_atom ? sc.seg_of_atom[row - 1] : sc.frame_n ? (int)((row - 1) / sc.frame_n) : 0), 4);
}

// ------------------------------------------------------------------------------------------------ geometry
// pass 1: R = X[ids-1] - X[i] (ids-1 = -1 wraps to the last atom), D = |R|, block max -> atomic max of the
// float bit pattern (D >= 0). Output rows are shifted by the sink row.   src/model_operations.py:8-10
// Frames (blockIdx.y): F independent coordinate sets of Nf atoms sharing ONE ids table (the MD use of the reference,
// md_analysis/apply_model_md.ipynb cell 6: model(X_frame, ids_topk_of_frame0, q, M) per frame). Frame f becomes atoms
// f*Nf+1 .. (f+1)*Nf of the internal batch; the wrap target and the max are per frame, exactly as in F separate calls.
// Segments (seg_of_atom != nullptr, F = 1): a ragged batch whose structures must behave like separate calls (the reference's
// bulk loops run one structure per call, apply_model.ipynb:139-167): atom i belongs to structure seg_of_atom[i] whose atoms
// are [seg_end[s-1], seg_end[s]); padding wraps to the structure's own last atom and the max is per structure.
template <typename IdT>
__global__ __launch_bounds__(256) void k_unpack1(int Nf, int k, const float* __restrict__ X, int64_t xs_frame, int64_t xs_atom,
                                                 const IdT* __restrict__ ids, int* __restrict__ ids_s, float4* __restrict__ geo,
                                                 unsigned* __restrict__ dmax_bits, int* __restrict__ err_flag,
                                                 const int* __restrict__ seg_of_atom, const int* __restrict__ seg_end, SegBoundsArgs sb, int pad_cols) {
    // pad_cols: the columns a layer can gather from (the largest nn of the config): a zero id in one of THEM is a padded slot for the
    // conditioning trigger; zero padding behind them (a table of k < 64 columns under a model whose layers use at most k) is never read
    // (sb: the residue segments of the pool layer, model_operations.py:199-211, only depend on res_of_atom - the thread of an atom's
    //  first slot finds them here, one launch less at the end of the forward)
    float d = 0.0f;
    const int f = blockIdx.y;
    const float* Xf = X + (int64_t)f * xs_frame;
    const int64_t a0 = (int64_t)f * Nf;
    if (seg_of_atom) {   // ragged structures: one thread per slot, one integer atomic per wave and structure touched
        const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
        if (e >= (int64_t)Nf * KMAX) return;
        const int i = (int)(e >> 6), c = (int)(e & 63);
        if (sb.roa && c == 0) seg_bound_atom(i, sb.R, sb.roa, sb.lo_enc, sb.hi, sb.err_flag);
        const int sg = seg_of_atom[i];
        long long id = c < k ? (long long)ids[(size_t)i * k + c] : 0;
        if (id < 0 || id > Nf) { atomicOr(err_flag, 1); id = 0; }
        if (__ballot(id == 0 && c < pad_cols) != 0 && (threadIdx.x & 63) == 0) pad_flag_at(err_flag, (int64_t)i + 1);      // (a wave = the 64 slots of one atom)
        const long long j = id > 0 ? id - 1 : (long long)seg_end[sg] - 1;
        const float* xj = Xf + j * xs_atom;
        const float* xi = Xf + (int64_t)i * xs_atom;
        const float rx = xj[0] - xi[0], ry = xj[1] - xi[1], rz = xj[2] - xi[2];
        const float dd = sqrtf(fmaf(rz, rz, fmaf(ry, ry, rx * rx)));      // torch.norm's FMA chain (knn_key)
        ids_s[(size_t)(i + 1) * KMAX + c] = (int)id;
        geo[(size_t)(i + 1) * KMAX + c] = make_float4(rx, ry, rz, dd);
        d = dd;
        for (int off = 32; off > 0; off >>= 1) d = fmaxf(d, __shfl_xor(d, off));   // a wave = the 64 slots of ONE atom
        // the maximum only grows: a wave whose value does not beat what is already published (L2-served read) skips the atomic,
        // so a structure costs a few hundred atomics on its word instead of one per atom
        if ((threadIdx.x & 63) == 0 && d > 0.0f &&
            __float_as_uint(d) > __hip_atomic_load(dmax_bits + sg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(dmax_bits + sg, __float_as_uint(d));
        return;
    }
    // 4 slots per thread (grid-stride by the grid size) keeps the number of blocks, hence atomics, at a quarter
    bool padded = false;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < (int64_t)Nf * KMAX; e += (int64_t)gridDim.x * 256) {
        const int i = (int)(e >> 6), c = (int)(e & 63);
        if (sb.roa && c == 0) seg_bound_atom(i, sb.R, sb.roa, sb.lo_enc, sb.hi, sb.err_flag);
        long long id = c < k ? (long long)ids[(size_t)i * k + c] : 0;
        if (id < 0 || id > Nf) { atomicOr(err_flag, 1); id = 0; }
        padded = padded || (id == 0 && c < pad_cols);
        long long j = id - 1;
        if (j < 0) j += Nf;
        const float* xj = Xf + j * xs_atom;
        const float* xi = Xf + (int64_t)i * xs_atom;
        const float rx = xj[0] - xi[0], ry = xj[1] - xi[1], rz = xj[2] - xi[2];
        const float dd = sqrtf(fmaf(rz, rz, fmaf(ry, ry, rx * rx)));      // torch.norm's FMA chain (knn_key)
        d = fmaxf(d, dd);
        ids_s[(size_t)(a0 + i + 1) * KMAX + c] = id ? (int)(id + a0) : 0;
        geo[(size_t)(a0 + i + 1) * KMAX + c] = make_float4(rx, ry, rz, dd);
    }
    // wave max -> block max -> ONE atomic per block (a single word saturates at ~90 atomics/us: one per wave
    // cost 277 us at 24k atoms)
    if (__ballot(padded) != 0 && (threadIdx.x & 63) == 0) pad_flag_at(err_flag, a0 + 1);      // (a block = atoms of ONE frame / of the one collated call)
    __shared__ float wmax[4];
    for (int off = 32; off > 0; off >>= 1) d = fmaxf(d, __shfl_xor(d, off));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
        if (m > 0.0f) atomicMax(dmax_bits + f, __float_as_uint(m));
    }
}

// pass 2: D += max(D) * (D < 1e-2);  R /= D.  Also writes the sink row 0.   src/model_operations.py:12-20
__global__ __launch_bounds__(256) void k_unpack2(int N, int Nf, int* __restrict__ ids_s, float4* __restrict__ geo,
                                                 const unsigned* __restrict__ dmax_bits, const int* __restrict__ seg_of_atom) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;   // over (N+1) * 64 slots
    if (e >= (int64_t)(N + 1) * KMAX) return;
    if (e < KMAX) { ids_s[e] = 0; geo[e] = make_float4(0.f, 0.f, 0.f, 0.f); return; }
    const float dmax = __uint_as_float(dmax_bits[seg_of_atom ? seg_of_atom[(e >> 6) - 1] : ((e >> 6) - 1) / Nf]);
    float4 g = geo[e];
    const float d = g.w + dmax * (g.w < 1e-2f ? 1.0f : 0.0f);
    geo[e] = make_float4(g.x / d, g.y / d, g.z / d, d);
}

// residue column of every atom of an F-frame batch: roa_f[f*Nf + i] = roa[i] + f*R
__global__ __launch_bounds__(256) void k_expand_roa(int Nf, int R, int F, const int* __restrict__ roa, int* __restrict__ roa_f,
                                                    int* __restrict__ err_flag) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)Nf * F) return;
    const int f = (int)(e / Nf), i = (int)(e % Nf);
    int r = roa[i];
    if (r < 0 || r >= R) { atomicOr(err_flag, 2); r = 0; }
    roa_f[e] = r + f * R;
}

// ------------------------------------------------------------------------------------------------ layer v1
// One workgroup = 64 edge rows = A = 64/NN centre atoms. fp32 VALU, operands staged k-major in LDS so every
// thread owns one output column and 32 rows (broadcast ds_read_b128 of the rows, coalesced weight reads).
// The centre block X_n(i) of the 193-wide edge input is folded into a per-centre partial sum (exact algebra).
constexpr int LD = 68;   // row stride of the k-major LDS tiles: 64 rows + 4 pad (keeps b128 alignment)

struct LayerSmem {
    float xe[129 * LD];    // varying part of X_e, k-major: 0 d | 1..32 q_j | 33..64 |p_j| | 65..96 p_i.r | 97..128 p_j.r ; reused as h2[128][LD]
    float h1[128 * LD];    // layer-1 activations, k-major; reused as kv[76][LD] (0-2 Kq, 3-11 Kp, 12-43 V0, 44-75 V1)
    float xn[8 * 64];      // centre node features [q_i | |p_i|]
    float pis[8 * 96];     // centre p_i
    float cpart[8 * 128];  // b1 + W1[:, 1:65] X_n(i)
    float4 geo[64];        // (r_hat, d) per row
    int nb[64];            // neighbour id per row
    float tmp[8 * 64];
    float Q[8 * 16];
    float lg[8 * 64];      // [h][part][row] logits -> attention weights
    float ex[8 * 64];
    float zs[8 * 256];     // [centre][Zq | Zp_x | Zp_y | Zp_z][h*32+s]
};

template <int NN>
__global__ __launch_bounds__(256) void k_layer_v1(const float* __restrict__ W, LayerW lw, int N1,
                                                  const int* __restrict__ ids_s, const float4* __restrict__ geo,
                                                  const float* __restrict__ q_in, const float* __restrict__ p_in,
                                                  float* __restrict__ q_out, float* __restrict__ p_out) {
    constexpr int A = 64 / NN;
    __shared__ LayerSmem sm;
    const int t = threadIdx.x;
    const int c0 = blockIdx.x * A;
    const float sdk = sqrtf((float)NK);

    // ---- phase 0: rows' neighbour ids + geometry, centre states
    if (t < 64) {
        const int a = t / NN, c = t % NN, i = c0 + a;
        const bool valid = i < N1;
        sm.nb[t] = valid ? ids_s[(size_t)i * KMAX + c] : 0;
        sm.geo[t] = valid ? geo[(size_t)i * KMAX + c] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    {
        const int a = t >> 5, s = t & 31;
        if (a < A) {
            const int i = min(c0 + a, N1 - 1);
            const float qv = q_in[(size_t)i * S + s];
            const float p0 = p_in[(size_t)i * 96 + s], p1 = p_in[(size_t)i * 96 + 32 + s], p2 = p_in[(size_t)i * 96 + 64 + s];
            sm.xn[a * 64 + s] = qv;
            sm.xn[a * 64 + 32 + s] = sqrtf(p0 * p0 + p1 * p1 + p2 * p2);
            sm.pis[a * 96 + s] = p0; sm.pis[a * 96 + 32 + s] = p1; sm.pis[a * 96 + 64 + s] = p2;
        }
    }
    __syncthreads();

    // ---- phase 1: gather neighbour states, build the varying edge features (k-major)
    {
        const int s = t & 31, rg = t >> 5;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int r = rg + 8 * it, a = r / NN;
            const int j = sm.nb[r];
            const float4 g = sm.geo[r];
            const float qj = q_in[(size_t)j * S + s];
            const float pj0 = p_in[(size_t)j * 96 + s], pj1 = p_in[(size_t)j * 96 + 32 + s], pj2 = p_in[(size_t)j * 96 + 64 + s];
            sm.xe[(1 + s) * LD + r] = qj;
            sm.xe[(33 + s) * LD + r] = sqrtf(pj0 * pj0 + pj1 * pj1 + pj2 * pj2);
            sm.xe[(65 + s) * LD + r] = sm.pis[a * 96 + s] * g.x + sm.pis[a * 96 + 32 + s] * g.y + sm.pis[a * 96 + 64 + s] * g.z;
            sm.xe[(97 + s) * LD + r] = pj0 * g.x + pj1 * g.y + pj2 * g.z;
            if (s == 0) sm.xe[r] = g.w;
        }
    }
    // ---- phase 2: per-centre part of edge layer 1, and the node query MLP (first layer)
    const int o = t & 127, rh = t >> 7;
    for (int a = rh; a < A; a += 2) {
        float acc = W[lw.b1 + o];
        for (int k = 0; k < 64; ++k) acc += sm.xn[a * 64 + k] * W[lw.w1 + (1 + k) * 128 + o];
        sm.cpart[a * 128 + o] = acc;
    }
    {
        const int a = t >> 5, s = t & 31;
        if (a < A) sm.tmp[a * 64 + s] = elu(g32_linear(W, lw.nqm.l[0], sm.xn + a * 64, s));
    }
    __syncthreads();
    {
        const int a = t >> 5, s = t & 31;
        if (a < A) sm.tmp[a * 64 + 32 + s] = elu(g32_linear(W, lw.nqm.l[1], sm.tmp + a * 64, s));
    }
    __syncthreads();
    {
        const int a = t >> 5, s = t & 31;
        if (a < A && s < 12) sm.Q[a * 16 + s] = g32_linear(W, lw.nqm.l[2], sm.tmp + a * 64 + 32, s);
    }

    // ---- phase 3: edge layer 1 -> h1 (ELU), column o, rows rh*32..+31
    float acc[32];
    {
#pragma unroll
        for (int m = 0; m < 32; ++m) acc[m] = sm.cpart[((rh * 32 + m) / NN) * 128 + o];
        for (int k = 0; k < 129; ++k) {
            const float w = W[lw.w1 + (k == 0 ? 0 : 64 + k) * 128 + o];
            const float4* xr = reinterpret_cast<const float4*>(sm.xe + k * LD + rh * 32);
#pragma unroll
            for (int m4 = 0; m4 < 8; ++m4) {
                const float4 v = xr[m4];
                acc[4 * m4 + 0] += v.x * w; acc[4 * m4 + 1] += v.y * w; acc[4 * m4 + 2] += v.z * w; acc[4 * m4 + 3] += v.w * w;
            }
        }
        float4* hw = reinterpret_cast<float4*>(sm.h1 + o * LD + rh * 32);
#pragma unroll
        for (int m4 = 0; m4 < 8; ++m4)
            hw[m4] = make_float4(elu(acc[4 * m4]), elu(acc[4 * m4 + 1]), elu(acc[4 * m4 + 2]), elu(acc[4 * m4 + 3]));
    }
    __syncthreads();

    // ---- phase 4: edge layer 2 (block diagonal: eq 32->32, ep 32->32, ev 64->64) -> h2 (in the xe region)
    {
        int kb, kn, ldw, oc; const float* wp;
        if (o < 32) { kb = 0; kn = 32; ldw = 32; oc = o; wp = W + lw.w2eq; }
        else if (o < 64) { kb = 32; kn = 32; ldw = 32; oc = o - 32; wp = W + lw.w2ep; }
        else { kb = 64; kn = 64; ldw = 64; oc = o - 64; wp = W + lw.w2ev; }
        const float b = W[lw.b2 + o];
#pragma unroll
        for (int m = 0; m < 32; ++m) acc[m] = b;
        for (int k = 0; k < kn; ++k) {
            const float w = wp[k * ldw + oc];
            const float4* xr = reinterpret_cast<const float4*>(sm.h1 + (kb + k) * LD + rh * 32);
#pragma unroll
            for (int m4 = 0; m4 < 8; ++m4) {
                const float4 v = xr[m4];
                acc[4 * m4 + 0] += v.x * w; acc[4 * m4 + 1] += v.y * w; acc[4 * m4 + 2] += v.z * w; acc[4 * m4 + 3] += v.w * w;
            }
        }
        float4* hw = reinterpret_cast<float4*>(sm.xe + o * LD + rh * 32);
#pragma unroll
        for (int m4 = 0; m4 < 8; ++m4)
            hw[m4] = make_float4(elu(acc[4 * m4]), elu(acc[4 * m4 + 1]), elu(acc[4 * m4 + 2]), elu(acc[4 * m4 + 3]));
    }
    __syncthreads();

    // ---- phase 5: edge layer 3 -> kv (in the h1 region): rows 0-2 Kq, 3-11 Kp (raw, chunk t = rows 3+3t..), 12-75 V
    if (o < 76) {
        int kb, kn, ldw, oc; const float* wp;
        if (o < 3) { kb = 0; kn = 32; ldw = 3; oc = o; wp = W + lw.w3eq; }
        else if (o < 12) { kb = 32; kn = 32; ldw = 9; oc = o - 3; wp = W + lw.w3ep; }
        else { kb = 64; kn = 64; ldw = 64; oc = o - 12; wp = W + lw.w3ev; }
        const float b = W[lw.b3 + o];
#pragma unroll
        for (int m = 0; m < 32; ++m) acc[m] = b;
        for (int k = 0; k < kn; ++k) {
            const float w = wp[k * ldw + oc];
            const float4* xr = reinterpret_cast<const float4*>(sm.xe + (kb + k) * LD + rh * 32);
#pragma unroll
            for (int m4 = 0; m4 < 8; ++m4) {
                const float4 v = xr[m4];
                acc[4 * m4 + 0] += v.x * w; acc[4 * m4 + 1] += v.y * w; acc[4 * m4 + 2] += v.z * w; acc[4 * m4 + 3] += v.w * w;
            }
        }
        float4* hw = reinterpret_cast<float4*>(sm.h1 + o * LD + rh * 32);
#pragma unroll
        for (int m4 = 0; m4 < 8; ++m4) hw[m4] = make_float4(acc[4 * m4], acc[4 * m4 + 1], acc[4 * m4 + 2], acc[4 * m4 + 3]);
    }
    __syncthreads();
    const float* kv = sm.h1;

    // ---- phase 7: attention logits and softmax. part 0: scalar keys over NN slots; parts 1..3: the three
    // vector-key chunks, softmaxed TOGETHER over 3*NN slots (chunk-major, model_operations.py:125,140)
    {
        const int r = t & 63, part = t >> 6, a = r / NN, g0 = a * NN;
        const int kr = part == 0 ? 0 : 3 + (part - 1) * 3;
        float l[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float* Qv = sm.Q + a * 16 + (part ? 6 : 0) + h * 3;
            l[h] = (Qv[0] * kv[(kr + 0) * LD + r] + Qv[1] * kv[(kr + 1) * LD + r] + Qv[2] * kv[(kr + 2) * LD + r]) / sdk;
            sm.lg[(h * 4 + part) * 64 + r] = l[h];
        }
        __syncthreads();
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float mx = -INFINITY;
            if (part == 0) {
                for (int c = 0; c < NN; ++c) mx = fmaxf(mx, sm.lg[(h * 4) * 64 + g0 + c]);
            } else {
                for (int pp = 1; pp < 4; ++pp)
                    for (int c = 0; c < NN; ++c) mx = fmaxf(mx, sm.lg[(h * 4 + pp) * 64 + g0 + c]);
            }
            l[h] = expf(l[h] - mx);
            sm.ex[(h * 4 + part) * 64 + r] = l[h];
        }
        __syncthreads();
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float sum = 0.0f;
            if (part == 0) {
                for (int c = 0; c < NN; ++c) sum += sm.ex[(h * 4) * 64 + g0 + c];
            } else {
                for (int pp = 1; pp < 4; ++pp)
                    for (int c = 0; c < NN; ++c) sum += sm.ex[(h * 4 + pp) * 64 + g0 + c];
            }
            sm.lg[(h * 4 + part) * 64 + r] = l[h] / sum;
        }
        __syncthreads();
    }

    // ---- phase 8: attention-weighted sums Zq [h*32+s], Zp[x][h*32+s]   (model_operations.py:131-144)
    {
        const int s = t & 31, h = (t >> 5) & 1, xq = t >> 6;
        for (int a = 0; a < A; ++a) {
            const int g0 = a * NN;
            float z = 0.0f;
            if (xq == 0) {
                for (int c = 0; c < NN; ++c) z += sm.lg[(h * 4) * 64 + g0 + c] * kv[(12 + s) * LD + g0 + c];
            } else {
                const int x = xq - 1;
                float wsum = 0.0f, z3 = 0.0f;
                for (int c = 0; c < NN; ++c) {
                    const int r = g0 + c;
                    const float4 g = sm.geo[r];
                    const float gx = x == 0 ? g.x : (x == 1 ? g.y : g.z);
                    z += sm.lg[(h * 4 + 1) * 64 + r] * (kv[(44 + s) * LD + r] * gx);
                    wsum += sm.lg[(h * 4 + 2) * 64 + r];
                    z3 += sm.lg[(h * 4 + 3) * 64 + r] * p_in[(size_t)sm.nb[r] * 96 + x * 32 + s];
                }
                z += wsum * sm.pis[a * 96 + x * 32 + s];
                z += z3;
            }
            sm.zs[a * 256 + xq * 64 + h * 32 + s] = z;
        }
    }
    __syncthreads();

    // ---- phase 9: output MLPs + residual, sink row reset   (model_operations.py:147-152, :239-240)
    {
        const int a = t >> 5, s = t & 31;
        const bool act = a < A;
        if (act) sm.tmp[a * 64 + s] = elu(g32_linear(W, lw.qpm.l[0], sm.zs + a * 256, s));
        __syncthreads();
        if (act) sm.tmp[a * 64 + 32 + s] = elu(g32_linear(W, lw.qpm.l[1], sm.tmp + a * 64, s));
        __syncthreads();
        if (act) {
            const int i = c0 + a;
            if (i < N1) {
                const float qh = g32_linear(W, lw.qpm.l[2], sm.tmp + a * 64 + 32, s);
                const bool sink = i == 0;
                q_out[(size_t)i * S + s] = sink ? 0.0f : sm.xn[a * 64 + s] + qh;
#pragma unroll
                for (int x = 0; x < 3; ++x) {
                    const float ph = g32_linear(W, lw.ppm, sm.zs + a * 256 + (1 + x) * 64, s);
                    p_out[(size_t)i * 96 + x * 32 + s] = sink ? 0.0f : sm.pis[a * 96 + x * 32 + s] + ph;
                }
            }
        }
    }
}

void launch_layer_v1(hipStream_t st, const float* W, const LayerW& lw, int N1, const int* ids_s, const float4* geo,
                     const float* q_in, const float* p_in, float* q_out, float* p_out) {
    const int A = 64 / lw.nn;
    const dim3 grid((N1 + A - 1) / A), block(256);
    switch (lw.nn) {
        case 8: hipLaunchKernelGGL(k_layer_v1<8>, grid, block, 0, st, W, lw, N1, ids_s, geo, q_in, p_in, q_out, p_out); break;
        case 16: hipLaunchKernelGGL(k_layer_v1<16>, grid, block, 0, st, W, lw, N1, ids_s, geo, q_in, p_in, q_out, p_out); break;
        case 32: hipLaunchKernelGGL(k_layer_v1<32>, grid, block, 0, st, W, lw, N1, ids_s, geo, q_in, p_in, q_out, p_out); break;
        default: hipLaunchKernelGGL(k_layer_v1<64>, grid, block, 0, st, W, lw, N1, ids_s, geo, q_in, p_in, q_out, p_out); break;
    }
}

// ------------------------------------------------------------------------------------------------ residue pool
__global__ void k_seg_bounds(int N, int R, const int* __restrict__ roa, int* __restrict__ lo_enc, int* __restrict__ hi,
                             int* __restrict__ err_flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) seg_bound_atom(i, R, roa, lo_enc, hi, err_flag);
}

// per-atom pool logits a[i][0..7] = sam([q_i | |p_i|]) + F_member   (model_operations.py:199-205)
__global__ __launch_bounds__(256) void k_pool_logits(const float* __restrict__ W, MlpW sam, int N,
                                                     const float* __restrict__ q, const float* __restrict__ p,
                                                     float* __restrict__ a_out) {
    __shared__ __attribute__((aligned(16))) float zin[8][64];
    __shared__ __attribute__((aligned(16))) float hs[8][64];
    const int g = threadIdx.x >> 5, s = threadIdx.x & 31;
    const int i = blockIdx.x * 8 + g;
    const int ic = i < N ? i : N - 1;
    const float p0 = p[(size_t)ic * 96 + s], p1 = p[(size_t)ic * 96 + 32 + s], p2 = p[(size_t)ic * 96 + 64 + s];
    zin[g][s] = q[(size_t)ic * S + s];
    zin[g][32 + s] = sqrtf(p0 * p0 + p1 * p1 + p2 * p2);
    __syncthreads();
    hs[g][s] = elu(g32_linear(W, sam.l[0], zin[g], s));
    __syncthreads();
    hs[g][32 + s] = elu(g32_linear(W, sam.l[1], hs[g], s));
    __syncthreads();
    const float f_member = (1.0f - 1.0f + 1e-6f) / (1.0f - 1e-6f);
    if (i < N && s < 2 * PH) a_out[(size_t)i * 8 + s] = g32_linear(W, sam.l[2], hs[g] + 32, s) + f_member;
}

// one wave per residue: segmented softmax over its atoms for the 8 channels (2h = scalar head h, 2h+1 = vector
// head h), weighted sums flattened s*4+h, zdm / zdm_vec, norm, decoder.   model_operations.py:205-211, model.py:49-50
__global__ __launch_bounds__(64) void k_pool_reduce(const float* __restrict__ W, ModelW mw, int n_out, int R,
                                                    const float* __restrict__ q, const float* __restrict__ p,
                                                    const float* __restrict__ a, const int* __restrict__ roa,
                                                    const int* __restrict__ lo, const int* __restrict__ hi,
                                                    float* __restrict__ qr_out, float* __restrict__ pr_out,
                                                    float* __restrict__ z_out, const int* __restrict__ flags, SatCtx sc, int only_flagged) {
    __shared__ __attribute__((aligned(16))) float qh[128];
    __shared__ __attribute__((aligned(16))) float ph[3][128];
    __shared__ __attribute__((aligned(16))) float hs[64];
    __shared__ __attribute__((aligned(16))) float zr[64];
    const int r = blockIdx.x;
    const int lane = threadIdx.x, s = lane & 31, hf = lane >> 5;
    const int i0 = 0x7fffffff - lo[r], i1 = hi[r];      // (lo is stored as 0x7fffffff - first member; an empty residue gives i0 > i1)
    // bad ids / residue columns: every logit NaN. An activation beyond the f16 range on the split-MFMA path (bit 2 of the word of the
    // residue's STRUCTURE): the result would be silently wrong (ELU maps the NaN of an overflowed product to 0), so it is made loud -
    // NaN for that structure's residues; the structures that stayed in range keep their logits
    const bool bad = i0 >= i1 || (*flags & 3);      // (empty residue: the reference degenerates to a whole-batch softmax; NaN here)
    bool ranged = false;
    if (!bad && sc.sflags) ranged = (sc.sflags[sc.seg_of_atom ? sc.seg_of_atom[i0] : sc.frame_n ? i0 / sc.frame_n : 0] & 4) != 0;
    if (only_flagged) {          // fp32 repeat: only the residues of the structures whose guard fired are (re)written
        if (bad || !ranged) return;
    } else if (bad || ranged) {
        if (lane < n_out) z_out[(size_t)r * n_out + lane] = __uint_as_float(0x7fc00000u);
        return;
    }
    float mx[4], den[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { mx[c] = -INFINITY; den[c] = 0.0f; }
#ifdef PESTO_ABL_POOL_NOSWEEP
    if (0)
#endif
    for (int i = i0; i < i1; ++i)
        if (roa[i] == r) {
            const float4 v = *reinterpret_cast<const float4*>(a + (size_t)i * 8 + 4 * hf);
            mx[0] = fmaxf(mx[0], v.x); mx[1] = fmaxf(mx[1], v.y); mx[2] = fmaxf(mx[2], v.z); mx[3] = fmaxf(mx[3], v.w);
        }
#ifdef PESTO_ABL_POOL_NOSWEEP
    if (0)
#endif
    for (int i = i0; i < i1; ++i)
        if (roa[i] == r) {
            const float4 v = *reinterpret_cast<const float4*>(a + (size_t)i * 8 + 4 * hf);
            den[0] += expf(v.x - mx[0]); den[1] += expf(v.y - mx[1]); den[2] += expf(v.z - mx[2]); den[3] += expf(v.w - mx[3]);
        }
    float aq[2] = {0.f, 0.f}, ap[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
#ifdef PESTO_ABL_POOL_NOSWEEP
    aq[0] = q[(size_t)i0 * S + s]; ap[0][0] = p[(size_t)i0 * 96 + s];
    if (0)
#endif
    for (int i = i0; i < i1; ++i)
        if (roa[i] == r) {
            const float4 v = *reinterpret_cast<const float4*>(a + (size_t)i * 8 + 4 * hf);
            const float w[4] = {expf(v.x - mx[0]) / den[0], expf(v.y - mx[1]) / den[1], expf(v.z - mx[2]) / den[2], expf(v.w - mx[3]) / den[3]};
            const float qv = q[(size_t)i * S + s];
            const float pv[3] = {p[(size_t)i * 96 + s], p[(size_t)i * 96 + 32 + s], p[(size_t)i * 96 + 64 + s]};
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                aq[hh] += qv * w[2 * hh];
#pragma unroll
                for (int x = 0; x < 3; ++x) ap[hh][x] += pv[x] * w[2 * hh + 1];
            }
        }
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int h = 2 * hf + hh;
        qh[s * PH + h] = aq[hh];
#pragma unroll
        for (int x = 0; x < 3; ++x) ph[x][s * PH + h] = ap[hh][x];
    }
    __syncthreads();
#ifdef PESTO_ABL_POOL_NOLINEAR
    if (hf == 0 && s < n_out) z_out[(size_t)r * n_out + s] = qh[s] + ph[0][s];
    return;
#endif
    // zdm on lanes 0..31, zdm_vec on all lanes (lane>>5 picks x = 0/1, then x = 2 by the first half)
    float v = 0.0f;
    if (hf == 0) hs[s] = elu(g32_linear(W, mw.zdm.l[0], qh, s));
    __syncthreads();
    if (hf == 0) v = elu(g32_linear(W, mw.zdm.l[1], hs, s));
    __syncthreads();
    if (hf == 0) hs[32 + s] = v;
    __syncthreads();
    float qrv = 0.0f;
    if (hf == 0) qrv = g32_linear(W, mw.zdm.l[2], hs + 32, s);
    const float pr0 = g32_linear(W, mw.zdm_vec, ph[hf], s);              // x = hf (0 or 1)
    const float pr2 = hf == 0 ? g32_linear(W, mw.zdm_vec, ph[2], s) : 0.0f;
    // gather the three components on the first half to take the norm
    const float pr1 = __shfl(pr0, 32 + s);
    if (hf == 0) {
        zr[s] = qrv;
        zr[32 + s] = sqrtf(pr0 * pr0 + pr1 * pr1 + pr2 * pr2);
        if (qr_out) qr_out[(size_t)r * S + s] = qrv;
        if (pr_out) { pr_out[(size_t)r * 96 + s] = pr0; pr_out[(size_t)r * 96 + 32 + s] = pr1; pr_out[(size_t)r * 96 + 64 + s] = pr2; }
    }
    __syncthreads();
    if (mw.dm.depth == 3) {
        if (hf == 0) hs[s] = elu(g32_linear(W, mw.dm.l[0], zr, s));
        __syncthreads();
        if (hf == 0) v = elu(g32_linear(W, mw.dm.l[1], hs, s));
        __syncthreads();
        if (hf == 0) hs[32 + s] = v;
        __syncthreads();
        if (hf == 0 && s < n_out) z_out[(size_t)r * n_out + s] = g32_linear(W, mw.dm.l[2], hs + 32, s);
    } else {
        if (hf == 0 && s < n_out) z_out[(size_t)r * n_out + s] = g32_linear(W, mw.dm.l[0], zr, s);
    }
}

// ------------------------------------------------------------------------------------------------ launchers
void launch_embed(hipStream_t st, const float* W, const MlpW& em, int N, int nq, int n0, const float* q0, float* q_state, float* p_zero,
                  ClearArgs clr, SatCtx sc) {
    hipLaunchKernelGGL(k_embed, dim3((N + 7) / 8), dim3(256), 0, st, W, em, N, nq, n0, q0, q_state, p_zero, clr, sc);
}

void launch_unpack(hipStream_t st, int Nf, int F, int k, const float* X, int64_t xs_frame, int64_t xs_atom, const void* ids,
                   int ids_kind, int* ids_s, float4* geo, unsigned* dmax_bits, int* err_flag, const int* seg_of_atom, const int* seg_end,
                   SegBoundsArgs sb, bool skip_pass2, int pad_cols) {
    const int64_t n1 = (int64_t)Nf * KMAX, nall = (int64_t)Nf * F * KMAX;
    const dim3 grid1((unsigned)((n1 + (seg_of_atom ? 255 : 1023)) / (seg_of_atom ? 256 : 1024)), (unsigned)F),
        grid2((unsigned)((nall + KMAX + 255) / 256));
    if (ids_kind == PESTO_IDS_INT64)
        hipLaunchKernelGGL(k_unpack1<long long>, grid1, dim3(256), 0, st, Nf, k, X, xs_frame, xs_atom, (const long long*)ids, ids_s, geo,
                           dmax_bits, err_flag, seg_of_atom, seg_end, sb, pad_cols);
    else
        hipLaunchKernelGGL(k_unpack1<int>, grid1, dim3(256), 0, st, Nf, k, X, xs_frame, xs_atom, (const int*)ids, ids_s, geo, dmax_bits,
                           err_flag, seg_of_atom, seg_end, sb, pad_cols);
    if (!skip_pass2) hipLaunchKernelGGL(k_unpack2, grid2, dim3(256), 0, st, Nf * F, Nf, ids_s, geo, dmax_bits, seg_of_atom);      // (else: inside the node launch, launch_node)
}

void launch_expand_roa(hipStream_t st, int Nf, int R, int F, const int* roa, int* roa_f, int* err_flag) {
    hipLaunchKernelGGL(k_expand_roa, dim3((unsigned)(((int64_t)Nf * F + 255) / 256)), dim3(256), 0, st, Nf, R, F, roa, roa_f, err_flag);
}

void launch_pool(hipStream_t st, const float* W, const ModelW& mw, int n_out, int N, int R, const float* q, const float* p,
                 const int* roa, float* a_tmp, int* lo, int* hi, int* err_flag, float* qr_out, float* pr_out, float* z_out, bool bounds_ready,
                 SatCtx sc, bool only_flagged) {
    if (!bounds_ready) {      // (the forward has them from its embed launch: lo / hi cleared with the other per-forward words)
        (void)hipMemsetAsync(lo, 0, (size_t)R * sizeof(int), st);
        (void)hipMemsetAsync(hi, 0, (size_t)R * sizeof(int), st);
        hipLaunchKernelGGL(k_seg_bounds, dim3((N + 255) / 256), dim3(256), 0, st, N, R, roa, lo, hi, err_flag);
    }
    hipLaunchKernelGGL(k_pool_logits, dim3((N + 7) / 8), dim3(256), 0, st, W, mw.sam, N, q, p, a_tmp);
    hipLaunchKernelGGL(k_pool_reduce, dim3(R), dim3(64), 0, st, W, mw, n_out, R, q, p, a_tmp, roa, lo, hi, qr_out, pr_out, z_out, err_flag, sc, only_flagged ? 1 : 0);
}


// ------------------------------------------------------------------------------------------------ k-NN topology + collate
// replaces extract_topology (src/data_encoding.py:87-102, dense [N,N,3] on the host) and the index part of
// collate_batch_features (src/dataset.py:100-109) with one kernel over a concatenated batch.
// Contract reproduced exactly: D = |X_j - X_i| in fp32 (products and sums rounded separately, like torch.norm over a
// 3-vector), entries with D < 1e-2 (self, coincident atoms) are pushed behind every other atom of the structure - the
// reference adds max(D) to them, so they sort after all unmasked entries and by D among themselves - and the
// min(k, N_s) smallest keys are emitted in ascending order as 1-based batch-global ids, zero padded to KMAX columns.
// One wave per query atom: candidates stream through in chunks of 64 (one per lane); the running k best are kept one
// per lane, sorted across the wave; a chunk that cannot improve the current worst is rejected with one ballot, the
// others are bitonic-sorted and merged (64-bit keys: [masked | distance bits | index], so ties break by index).
__device__ __forceinline__ unsigned long long shfl_xor64(unsigned long long v, int m) {
    const unsigned lo = __shfl_xor((unsigned)v, m), hi = __shfl_xor((unsigned)(v >> 32), m);
    return ((unsigned long long)hi << 32) | lo;
}
// one compare-exchange stage of a bitonic network across the wave: partner = lane ^ j, ascending if `up`
__device__ __forceinline__ unsigned long long bitonic_step(unsigned long long v, int lane, int j, bool up) {
    const unsigned long long o = shfl_xor64(v, j);
    const bool lower = (lane & j) == 0;
    const bool take_min = lower == up;
    return take_min ? (v < o ? v : o) : (v < o ? o : v);
}

// ---- cell grid for large structures (N_s >= KNN_CELL_MIN): brute force is O(N_s^2); with a uniform grid a query only meets the
// atoms of the (2s+1)^3 cells around it. Exactness is kept by a completeness test: every atom outside that block is at least
// s*h away from the query, so once the k-th best key is an unmasked distance strictly below s*h the k best are final; otherwise
// the block grows (s = 2, 3, 5, 8, ... until it covers the grid). Keys, tie-breaks and the merge network are those of the
// brute-force path, so both paths return identical tables.
constexpr int KNN_CELL_MIN = 1024;        // structures at least this large use the grid (measured: 2x at N = 3,000, 5.7x at N = 20,000)
constexpr int KNN_GRID_MAX = 32;          // cells per axis (h grows beyond 32 * 4.5 A = 144 A of extent)
constexpr int KNN_MAXC = KNN_GRID_MAX * KNN_GRID_MAX * KNN_GRID_MAX;
struct KnnGrid { float minx, miny, minz, h, inv_h; int nx, ny, nz, use, slot; };   // use = 0: brute force for this structure;
                                                                                  // slot = which block of the cell arrays it owns

__global__ __launch_bounds__(256) void k_grid_setup(int n_struct, const int* __restrict__ offsets, const int* __restrict__ slots,
                                                    const float* __restrict__ X, KnnGrid* __restrict__ grids, int* __restrict__ cell_cnt) {
    const int s = blockIdx.x;
    const int s0 = offsets[s], s1 = offsets[s + 1];
    __shared__ float red[6][256];
    __shared__ KnnGrid gsh;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    const bool big = slots[s] >= 0;
    if (big)
        for (int i = s0 + threadIdx.x; i < s1; i += 256)
#pragma unroll
            for (int c = 0; c < 3; ++c) { const float v = X[3 * (size_t)i + c]; mn[c] = fminf(mn[c], v); mx[c] = fmaxf(mx[c], v); }
#pragma unroll
    for (int c = 0; c < 3; ++c) { red[c][threadIdx.x] = mn[c]; red[3 + c][threadIdx.x] = mx[c]; }
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                red[c][threadIdx.x] = fminf(red[c][threadIdx.x], red[c][threadIdx.x + off]);
                red[3 + c][threadIdx.x] = fmaxf(red[3 + c][threadIdx.x], red[3 + c][threadIdx.x + off]);
            }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        KnnGrid g;
        g.use = 0; g.slot = slots[s]; g.nx = g.ny = g.nz = 0; g.minx = g.miny = g.minz = 0.f; g.h = g.inv_h = 0.f;
        if (big) {
            const float ex = red[3][0] - red[0][0], ey = red[4][0] - red[1][0], ez = red[5][0] - red[2][0];
            const float ext = fmaxf(ex, fmaxf(ey, ez));
            if (ext == ext && ext < 1e30f) {          // finite coordinates only; otherwise stay on the brute-force path
                g.h = fmaxf(4.5f, ext / (float)KNN_GRID_MAX * 1.0001f);
                g.inv_h = 1.0f / g.h;
                g.minx = red[0][0]; g.miny = red[1][0]; g.minz = red[2][0];
                g.nx = min(KNN_GRID_MAX, (int)(ex * g.inv_h) + 1); g.ny = min(KNN_GRID_MAX, (int)(ey * g.inv_h) + 1);
                g.nz = min(KNN_GRID_MAX, (int)(ez * g.inv_h) + 1);
                g.use = 1;
            }
        }
        grids[s] = g;
        gsh = g;
    }
    __syncthreads();
    if (gsh.use) {
        const int nc = gsh.nx * gsh.ny * gsh.nz;
        for (int c = threadIdx.x; c <= nc; c += 256) cell_cnt[(size_t)gsh.slot * (KNN_MAXC + 1) + c] = 0;
    }
}

__device__ __forceinline__ int knn_cell_of(const KnnGrid& g, float x, float y, float z) {
    const int cx = min(g.nx - 1, max(0, (int)((x - g.minx) * g.inv_h)));
    const int cy = min(g.ny - 1, max(0, (int)((y - g.miny) * g.inv_h)));
    const int cz = min(g.nz - 1, max(0, (int)((z - g.minz) * g.inv_h)));
    return (cz * g.ny + cy) * g.nx + cx;
}
__device__ __forceinline__ int knn_struct_of(int i, int n_struct, const int* __restrict__ offsets) {
    int lo = 0, hi = n_struct;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (offsets[mid] <= i) lo = mid; else hi = mid; }
    return lo;
}

__global__ __launch_bounds__(256) void k_grid_count(int n_total, int n_struct, const int* __restrict__ offsets, const float* __restrict__ X,
                                                    const KnnGrid* __restrict__ grids, int* __restrict__ cell_cnt, int* __restrict__ cell_of) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_total) return;
    const int s = knn_struct_of(i, n_struct, offsets);
    const KnnGrid g = grids[s];
    if (!g.use) return;
    const int c = knn_cell_of(g, X[3 * (size_t)i], X[3 * (size_t)i + 1], X[3 * (size_t)i + 2]);
    cell_of[i] = c;
    atomicAdd(&cell_cnt[(size_t)g.slot * (KNN_MAXC + 1) + c], 1);
}

// exclusive scan of one structure's cell counts (one workgroup per structure; <= 32,768 cells) -> cell_start, cursor copy
__global__ __launch_bounds__(1024) void k_grid_scan(const KnnGrid* __restrict__ grids, int* __restrict__ cell_cnt, int* __restrict__ cell_cur) {
    const int s = blockIdx.x;
    const KnnGrid g = grids[s];
    if (!g.use) return;
    const int nc = g.nx * g.ny * g.nz;
    int* cnt = cell_cnt + (size_t)g.slot * (KNN_MAXC + 1);
    int* cur = cell_cur + (size_t)g.slot * (KNN_MAXC + 1);
    __shared__ int part[1024];
    const int per = (nc + 1023) / 1024;
    const int c0 = threadIdx.x * per, c1 = min(nc, c0 + per);
    int sum = 0;
    for (int c = c0; c < c1; ++c) sum += cnt[c];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {          // inclusive scan of the per-thread sums
        const int v = (int)threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = part[threadIdx.x] - sum;
    for (int c = c0; c < c1; ++c) { const int n = cnt[c]; cnt[c] = run; cur[c] = run; run += n; }
    if (threadIdx.x == 1023) cnt[nc] = part[1023];
}

// atoms in cell order: (x, y, z, local index) so that a candidate is one 16-byte load
__global__ __launch_bounds__(256) void k_grid_scatter(int n_total, int n_struct, const int* __restrict__ offsets, const float* __restrict__ X,
                                                      const KnnGrid* __restrict__ grids, const int* __restrict__ cell_of,
                                                      int* __restrict__ cell_cur, float4* __restrict__ sorted) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_total) return;
    const int s = knn_struct_of(i, n_struct, offsets);
    if (!grids[s].use) return;
    const int pos = atomicAdd(&cell_cur[(size_t)grids[s].slot * (KNN_MAXC + 1) + cell_of[i]], 1);
    sorted[offsets[s] + pos] = make_float4(X[3 * (size_t)i], X[3 * (size_t)i + 1], X[3 * (size_t)i + 2], __int_as_float(i - offsets[s]));
}

// merge one chunk of 64 candidate keys (one per lane, WORST = none) into the running top-64 (ascending across lanes)
__device__ __forceinline__ unsigned long long knn_merge(unsigned long long top, unsigned long long key, int lane) {
    const unsigned long long worst = ((unsigned long long)__shfl((unsigned)(top >> 32), 63) << 32) | __shfl((unsigned)top, 63);
    if (__ballot(key < worst) == 0ull) return top;        // nothing in this chunk beats the current k-th best
    for (int size = 2; size <= 64; size <<= 1)
        for (int jj = size >> 1; jj > 0; jj >>= 1) key = bitonic_step(key, lane, jj, (lane & size) == 0 || size == 64);
    // 64 smallest of (top ascending, chunk ascending): min(top[l], chunk[63-l]) is bitonic; merge it ascending
    const unsigned long long rev = shfl_xor64(key, 63);
    unsigned long long m = top < rev ? top : rev;
    for (int jj = 32; jj > 0; jj >>= 1) m = bitonic_step(m, lane, jj, true);
    return m;
}
__device__ __forceinline__ unsigned long long knn_key(float rx, float ry, float rz, int local_index) {
    // torch.norm over the contiguous xyz axis is the scalar loop acc = acc + v * v, which the reference's build contracts into an FMA
    // chain: d = sqrt(fma(z, z, fma(y, y, x * x))) reproduces its distance matrix bit for bit (555,000 pairs of a pdbs_test chain;
    // separately rounded products agree on 89 % only, and a one-ulp difference reorders two neighbours in about one row of 40,000)
    // (sqrtf, not __fsqrt_rn: the intrinsic lowers to the bare v_sqrt_f32 - one ulp off in places - while sqrtf is correctly rounded)
    const float d = sqrtf(__fmaf_rn(rz, rz, __fmaf_rn(ry, ry, __fmul_rn(rx, rx))));
    const unsigned masked = d < 1e-2f ? 1u : 0u;
    return ((unsigned long long)((masked << 31) | __float_as_uint(d)) << 32) | (unsigned)local_index;   // d >= 0: bit 31 is free
}

// Exact fp32 distance ties that STRADDLE a neighbourhood cut-off: the table orders equal distances by index, the reference's torch.topk
// leaves their order undefined (src/data_encoding.py:98-99), so for such a row the two disagree on which of two equally distant atoms
// belongs to the first 8 / 16 / 32 / 64 neighbours - the one difference between the tables that can move a logit (a tie inside a
// neighbourhood only reorders a sum). One wave per row: the keys (masked flag | distance bits, as knn_key) of the table's last column
// inside every cut, the number of columns inside the cut that carry that key, and - one scan over the atoms of the structure - the number
// of atoms that do: more atoms than columns = the tie straddles the cut. flags[i]: bit 0 / 1 / 2 / 3 = cut after column 8 / 16 / 32 / 64.
template <typename IdT>
__global__ __launch_bounds__(256) void k_knn_ties(int n_total, int n_struct, const int* __restrict__ offsets, const float* __restrict__ X, int k,
                                                  const IdT* __restrict__ ids, unsigned char* __restrict__ flags) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n_total) return;
    const int s = knn_struct_of(i, n_struct, offsets);
    const int s0 = offsets[s], s1 = offsets[s + 1];
    const float xi = X[3 * (size_t)i], yi = X[3 * (size_t)i + 1], zi = X[3 * (size_t)i + 2];
    auto dkey = [&](int j) -> unsigned {      // distance key of atom j (batch-global, 0-based): knn_key without the index
        return (unsigned)(knn_key(X[3 * (size_t)j] - xi, X[3 * (size_t)j + 1] - yi, X[3 * (size_t)j + 2] - zi, 0) >> 32);
    };
    // this lane's column of the table (k <= 64 columns; 0 = padding)
    const long long id = lane < k ? (long long)ids[(size_t)i * KMAX + lane] : 0;
    const bool have = id > 0;
    const unsigned mykey = have ? dkey((int)id - 1) : 0xffffffffu;
    // the four cut keys first, then ONE scan over the structure's atoms counting all four (round 4 scanned once per cut). The atom itself is
    // left out on both sides - as a column (structures of <= 64 atoms list it, masked, at the end) and in the scan: its own distance-0 key
    // used to count as a tie with every coincident atom and over-reported rows of structures with duplicated coordinates.
    unsigned kc[4];
    int inside[4], all[4] = {0, 0, 0, 0};
    bool live[4];
    const bool mine = have && (int)id - 1 != i;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int cut = 8 << c;
        live[c] = cut <= k && ids[(size_t)i * KMAX + min(cut, KMAX) - 1] > 0;       // fewer than `cut` neighbours: nothing beyond the cut
        kc[c] = __shfl(mykey, min(cut, 64) - 1);
        inside[c] = __popcll(__ballot(lane < cut && mine && mykey == kc[c]));
    }
    for (int j = s0 + lane; j < s1; j += 64) {
        const unsigned kj = j == i ? 0xfffffffeu : dkey(j);      // (no key equals this one: the masked flag is the top bit, the rest distance bits)
#pragma unroll
        for (int c = 0; c < 4; ++c) all[c] += kj == kc[c] ? 1 : 0;
    }
    unsigned out = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        int a = all[c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
        if (live[c] && a > inside[c]) out |= 1u << c;
    }
    if (lane == 0) flags[i] = (unsigned char)out;
}

template <typename IdT>
__global__ __launch_bounds__(256) void k_knn_collate(int n_total, int n_struct, const int* __restrict__ offsets,
                                                     const float* __restrict__ X, int k, IdT* __restrict__ ids_out,
                                                     const KnnGrid* __restrict__ grids, const int* __restrict__ cell_start,
                                                     const float4* __restrict__ sorted) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n_total) return;
    const int st = knn_struct_of(i, n_struct, offsets);      // wave-uniform
    const int s0 = offsets[st], s1 = offsets[st + 1];
    const float xi = X[3 * (size_t)i], yi = X[3 * (size_t)i + 1], zi = X[3 * (size_t)i + 2];
    const unsigned long long WORST = ~0ull;
    unsigned long long top = WORST;                 // lane l holds the (l+1)-th best key so far, ascending across lanes
    const int knn = min(k, s1 - s0);
    bool done = false;
    if (grids && grids[st].use) {
        const KnnGrid g = grids[st];
        const int* cs = cell_start + (size_t)g.slot * (KNN_MAXC + 1);
        const int cx = min(g.nx - 1, max(0, (int)((xi - g.minx) * g.inv_h)));
        const int cy = min(g.ny - 1, max(0, (int)((yi - g.miny) * g.inv_h)));
        const int cz = min(g.nz - 1, max(0, (int)((zi - g.minz) * g.inv_h)));
        for (int s = 2; !done; s = s < 3 ? 3 : (s * 8 + 4) / 5) {       // 2, 3, 5, 8, 13, ...
            top = WORST;
            const int x0 = max(0, cx - s), x1 = min(g.nx - 1, cx + s);
            for (int zc = max(0, cz - s); zc <= min(g.nz - 1, cz + s); ++zc)
                for (int yc = max(0, cy - s); yc <= min(g.ny - 1, cy + s); ++yc) {
                    const int row = (zc * g.ny + yc) * g.nx;
                    const int a0 = cs[row + x0], a1 = cs[row + x1 + 1];      // one contiguous run of the sorted atoms
                    for (int base = a0; base < a1; base += 64) {
                        unsigned long long key = WORST;
                        if (base + lane < a1) {
                            const float4 c = sorted[s0 + base + lane];
                            key = knn_key(c.x - xi, c.y - yi, c.z - zi, __float_as_int(c.w));
                        }
                        top = knn_merge(top, key, lane);
                    }
                }
            const bool whole = cx - s <= 0 && cy - s <= 0 && cz - s <= 0 && cx + s >= g.nx - 1 && cy + s >= g.ny - 1 && cz + s >= g.nz - 1;
            // k-th best: an unmasked distance strictly inside the radius the block guarantees (small margin for the rounding of
            // the cell assignment) -> no atom outside the block can enter or tie
            const unsigned long long kth = ((unsigned long long)__shfl((unsigned)(top >> 32), knn - 1) << 32) | __shfl((unsigned)top, knn - 1);
            const unsigned hi32 = (unsigned)(kth >> 32);
            const bool ok = kth != WORST && !(hi32 >> 31) && __uint_as_float(hi32) < (float)s * g.h * 0.9999f;
            done = whole || ok;
        }
    }
    if (!done) {
        top = WORST;
        for (int base = s0; base < s1; base += 64) {
            const int j = base + lane;
            unsigned long long key = WORST;
            if (j < s1) key = knn_key(X[3 * (size_t)j] - xi, X[3 * (size_t)j + 1] - yi, X[3 * (size_t)j + 2] - zi, j - s0);
            top = knn_merge(top, key, lane);
        }
    }
    // lanes 0..knn-1 hold the neighbours in ascending order; zero padding beyond (dataset.py:100,109)
    long long id = 0;
    if (lane < knn) id = (long long)(unsigned)(top & 0xffffffffu) + s0 + 1;
    if (lane < KMAX) ids_out[(size_t)i * KMAX + lane] = (IdT)id;
}

// ------------------------------------------------------------------------------------------------ device-side collate
// collate_batch_features (src/dataset.py:100-110) for per-structure arrays already copied to the device back to back:
// ids_out[i][c] = c < k_b ? ids_raw_b[i - off_b][c] + off_b + 1 : 0 (1-based batch-global, zero padded to KMAX columns) and
// roa_out[i] = roa_raw[i] + roff_b. meta[b] = {off_b, roff_b, N_b, R_b, k_b, idoff_b (element offset of the ragged ids)}.
struct CollateMeta { int off, roff, n, r, k; long long idoff; };
template <typename IdT>
__global__ __launch_bounds__(256) void k_collate(int n_total, int n_struct, const CollateMeta* __restrict__ meta, const IdT* __restrict__ ids_raw,
                                                 const int* __restrict__ roa_raw, int* __restrict__ ids_out, int* __restrict__ roa_out,
                                                 int* __restrict__ seg_of_atom, int* __restrict__ seg_end, int* __restrict__ err_flag) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e < n_struct) seg_end[e] = meta[e].off + meta[e].n;
    if (e >= (int64_t)n_total * KMAX) return;
    const int i = (int)(e >> 6), c = (int)(e & 63);
    int lo = 0, hi = n_struct;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (meta[mid].off <= i) lo = mid; else hi = mid; }
    const CollateMeta mb = meta[lo];
    long long id = 0;
    if (c < mb.k) {
        id = (long long)ids_raw[mb.idoff + (long long)(i - mb.off) * mb.k + c];
        if (id < 0 || id >= mb.n) { atomicOr(err_flag, 1); id = 0; } else id += mb.off + 1;
    }
    ids_out[e] = (int)id;
    if (c == 0) {
        int r = roa_raw[i];
        if (r < 0 || r >= mb.r) { atomicOr(err_flag, 2); r = 0; }
        roa_out[i] = r + mb.roff;
        seg_of_atom[i] = lo;
    }
}

// seg_of_atom[i] = index of the structure whose range [seg_end[s-1], seg_end[s]) holds atom i
__global__ __launch_bounds__(256) void k_segments(int n_total, int n_struct, const int* __restrict__ seg_end, int* __restrict__ seg_of_atom) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_total) return;
    int lo = 0, hi = n_struct - 1;          // first s with seg_end[s] > i
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (seg_end[mid] > i) hi = mid; else lo = mid + 1; }
    seg_of_atom[i] = lo;
}
void launch_segments(hipStream_t st, int n_total, int n_struct, const int* seg_end, int* seg_of_atom) {
    hipLaunchKernelGGL(k_segments, dim3((n_total + 255) / 256), dim3(256), 0, st, n_total, n_struct, seg_end, seg_of_atom);
}

void launch_collate(hipStream_t st, int n_total, int n_struct, const void* meta, const void* ids_raw, int ids_kind, const int* roa_raw,
                    int* ids_out, int* roa_out, int* seg_of_atom, int* seg_end, int* err_flag) {
    const dim3 grid((unsigned)(((int64_t)n_total * KMAX + 255) / 256)), block(256);
    if (ids_kind == PESTO_IDS_INT64)
        hipLaunchKernelGGL(k_collate<long long>, grid, block, 0, st, n_total, n_struct, (const CollateMeta*)meta, (const long long*)ids_raw, roa_raw,
                           ids_out, roa_out, seg_of_atom, seg_end, err_flag);
    else if (ids_kind == PESTO_IDS_UINT16)
        hipLaunchKernelGGL(k_collate<unsigned short>, grid, block, 0, st, n_total, n_struct, (const CollateMeta*)meta, (const unsigned short*)ids_raw,
                           roa_raw, ids_out, roa_out, seg_of_atom, seg_end, err_flag);
    else
        hipLaunchKernelGGL(k_collate<int>, grid, block, 0, st, n_total, n_struct, (const CollateMeta*)meta, (const int*)ids_raw, roa_raw, ids_out,
                           roa_out, seg_of_atom, seg_end, err_flag);
}

// One-hot features from their indices: q0[i][offs[c] + idx[i][c]] = 1 for the n_idx index columns of atom i (encode_features,
// src/data_encoding.py:78-84: element | residue name | atom name blocks), everything else 0. One thread per output float.
// An index outside its block sets the ids error bit (the forward then reports bad inputs).
__global__ __launch_bounds__(256) void k_onehot(int N, int n0, int n_idx, const unsigned char* __restrict__ idx, int o0, int o1, int o2,
                                                float* __restrict__ q0, int* __restrict__ err_flag) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)N * n0) return;
    const int i = (int)(e / n0), f = (int)(e - (int64_t)i * n0);
    const int offs[4] = {o0, o1, o2, n0};
    float v = 0.0f;
    for (int c = 0; c < n_idx; ++c) {
        const int hit = offs[c] + idx[(size_t)i * n_idx + c];
        if (hit >= (c + 1 < n_idx ? offs[c + 1] : n0)) atomicOr(err_flag, 1);
        if (hit == f) v = 1.0f;
    }
    q0[e] = v;
}
void launch_onehot(hipStream_t st, int N, int n0, int n_idx, const unsigned char* idx, const int* offs, float* q0, int* err_flag) {
    hipLaunchKernelGGL(k_onehot, dim3((unsigned)(((int64_t)N * n0 + 255) / 256)), dim3(256), 0, st, N, n0, n_idx, idx, offs[0], n_idx > 1 ? offs[1] : n0,
                       n_idx > 2 ? offs[2] : n0, q0, err_flag);
}

// ------------------------------------------------------------------------------------------------ post-processing
// p = sigmoid(z) per residue (apply_model.ipynb:160, interfaceome/apply_model.py:76) and its expansion to
// atoms, bf[c][i] = p[res_of_atom[i]][c] - what encode_bfactor (src/structure.py:208-218) does on the host with one
// numpy mask per residue. One thread per output element; z is tiny, so the atom part recomputes the sigmoid.
__global__ __launch_bounds__(256) void k_postprocess(int N, int R, int n_out, const float* __restrict__ z, const int* __restrict__ roa,
                                                     float* __restrict__ p_out, float* __restrict__ bf_out, int* __restrict__ err_flag) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t n_res = (int64_t)R * n_out;
    if (e < n_res) {
        if (p_out) p_out[e] = 1.0f / (1.0f + expf(-z[e]));
    } else if (bf_out && e < n_res + (int64_t)N * n_out) {
        const int64_t a = e - n_res;
        const int c = (int)(a / N), i = (int)(a % N);
        int r = roa[i];
        if (r < 0 || r >= R) { atomicOr(err_flag, 2); r = 0; }
        bf_out[a] = 1.0f / (1.0f + expf(-z[(size_t)r * n_out + c]));
    }
}

// Dense residue mask -> segments (Model.forward's M argument, model/model.py:32; one 1 per row by construction, src/data_encoding.py:73).
// HBM-read-bound: the mask of the headline batch is 288 MB (24,000 x 3,000 fp32), the answer 96 KB. One wave per row at a time, the
// lanes sweep the row's columns with 16-byte loads FOUR batches deep (4 x 1 KB of a row in flight per wave, eight waves per SIMD - a
// plain load - compare loop kept one load in flight and ran at 2.3 TB/s), count the members (> 0.5, as the host check does) and keep the
// column of the last one; a row with exactly one member gets its column, any other row -1. Workgroups stride over the rows (a grid of
// 2,048 workgroups instead of one per four rows: no tail of tiny workgroups). Every valid row marks its column in `seen` with the
// call's generation number (no clearing launch); k_mask_check poisons roa[0] when a column stayed empty. No host round trip: a
// poisoned entry fails the residue-column check of the forward that consumes the array.
typedef float mask_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void mask_scan4(const mask_f4 v, int c4, int& cnt, int& col) {
    if (v[0] > 0.5f) { ++cnt; col = 4 * c4; }
    if (v[1] > 0.5f) { ++cnt; col = 4 * c4 + 1; }
    if (v[2] > 0.5f) { ++cnt; col = 4 * c4 + 2; }
    if (v[3] > 0.5f) { ++cnt; col = 4 * c4 + 3; }
}
__global__ __launch_bounds__(256) void k_mask_to_segments(int N, int R, const float* __restrict__ M, int* __restrict__ roa, int* __restrict__ seen, int gen) {
    const int lane = threadIdx.x & 63;
    const int R4 = R >> 2;
    const mask_f4 zero4 = mask_f4{0.f, 0.f, 0.f, 0.f};
    for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < N; i += gridDim.x * 4) {
        const float* row = M + (size_t)i * R;
        int cnt = 0, col = -1;
        int c0 = 0;
        if ((((size_t)row) & 15) == 0) {
            const mask_f4* row4 = reinterpret_cast<const mask_f4*>(row);
            for (int c = lane; c < R4; c += 256) {
                // (streaming loads: the mask is read once and is larger than the Infinity Cache - it must not evict the weights, records and
                // states the layer kernels behind it work on)
                const mask_f4 v0 = __builtin_nontemporal_load(row4 + c);
                const mask_f4 v1 = c + 64 < R4 ? __builtin_nontemporal_load(row4 + c + 64) : zero4;
                const mask_f4 v2 = c + 128 < R4 ? __builtin_nontemporal_load(row4 + c + 128) : zero4;
                const mask_f4 v3 = c + 192 < R4 ? __builtin_nontemporal_load(row4 + c + 192) : zero4;
                mask_scan4(v0, c, cnt, col); mask_scan4(v1, c + 64, cnt, col); mask_scan4(v2, c + 128, cnt, col); mask_scan4(v3, c + 192, cnt, col);
            }
            c0 = R4 << 2;
        }
        for (int c = c0 + lane; c < R; c += 64)
            if (row[c] > 0.5f) { ++cnt; col = c; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            cnt += __shfl_xor(cnt, o);
            col = max(col, __shfl_xor(col, o));
        }
        if (lane == 0) {
            const bool ok = cnt == 1;
            roa[i] = ok ? col : -1;
            if (ok) seen[col] = gen;
        }
    }
}
__global__ __launch_bounds__(256) void k_mask_check(int R, const int* __restrict__ seen, int* __restrict__ roa, int gen) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < R && seen[r] != gen) roa[0] = -1;      // (benign race: every writer stores the same value)
}
// gen: a number that differs from the one of the previous call on the same `seen` buffer (and from what a fresh buffer holds)
void launch_mask_to_segments(hipStream_t st, int N, int R, const float* M, int* roa, int* seen, int gen) {
    const int rows4 = (N + 3) / 4;
    hipLaunchKernelGGL(k_mask_to_segments, dim3(rows4 < 2048 ? rows4 : 2048), dim3(256), 0, st, N, R, M, roa, seen, gen);
    hipLaunchKernelGGL(k_mask_check, dim3((R + 255) / 256), dim3(256), 0, st, R, seen, roa, gen);
}

void launch_postprocess(hipStream_t st, int N, int R, int n_out, const float* z, const int* roa, float* p_out, float* bf_out, int* err_flag) {
    const int64_t n = (int64_t)R * n_out + (bf_out ? (int64_t)N * n_out : 0);
    hipLaunchKernelGGL(k_postprocess, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, N, R, n_out, z, roa, p_out, bf_out, err_flag);
}

// grid buffers (all device): slots [n_struct] (block of the cell arrays a structure owns, -1 = brute force), grids [n_struct],
// cell_cnt / cell_cur [n_slots * (KNN_MAXC + 1)] ints, cell_of [n_total] ints, sorted [n_total] float4. use_grid = 0 skips the grid.
void launch_knn_collate(hipStream_t st, int n_total, int n_struct, const int* offsets, const float* X, int k, void* ids_out, int ids_kind,
                        int use_grid, const int* slots, void* grids, int* cell_cnt, int* cell_cur, int* cell_of, void* sorted) {
    if (use_grid) {
        hipLaunchKernelGGL(k_grid_setup, dim3(n_struct), dim3(256), 0, st, n_struct, offsets, slots, X, (KnnGrid*)grids, cell_cnt);
        hipLaunchKernelGGL(k_grid_count, dim3((n_total + 255) / 256), dim3(256), 0, st, n_total, n_struct, offsets, X, (const KnnGrid*)grids, cell_cnt, cell_of);
        hipLaunchKernelGGL(k_grid_scan, dim3(n_struct), dim3(1024), 0, st, (const KnnGrid*)grids, cell_cnt, cell_cur);
        hipLaunchKernelGGL(k_grid_scatter, dim3((n_total + 255) / 256), dim3(256), 0, st, n_total, n_struct, offsets, X, (const KnnGrid*)grids, cell_of, cell_cur,
                           (float4*)sorted);
    }
    const KnnGrid* gp = use_grid ? (const KnnGrid*)grids : nullptr;
    const dim3 grid((n_total + 3) / 4), block(256);
    if (ids_kind == PESTO_IDS_INT64)
        hipLaunchKernelGGL(k_knn_collate<long long>, grid, block, 0, st, n_total, n_struct, offsets, X, k, (long long*)ids_out, gp, cell_cnt, (const float4*)sorted);
    else
        hipLaunchKernelGGL(k_knn_collate<int>, grid, block, 0, st, n_total, n_struct, offsets, X, k, (int*)ids_out, gp, cell_cnt, (const float4*)sorted);
}

size_t knn_grid_struct_bytes() { return sizeof(KnnGrid); }
int knn_cell_min() { return KNN_CELL_MIN; }
int knn_cells_per_struct() { return KNN_MAXC + 1; }

void launch_knn_ties(hipStream_t st, int n_total, int n_struct, const int* offsets, const float* X, int k, const void* ids, int ids_kind,
                      unsigned char* flags) {
    const dim3 grid((n_total + 3) / 4), block(256);
    if (ids_kind == PESTO_IDS_INT64) hipLaunchKernelGGL(k_knn_ties<long long>, grid, block, 0, st, n_total, n_struct, offsets, X, k, (const long long*)ids, flags);
    else hipLaunchKernelGGL(k_knn_ties<int>, grid, block, 0, st, n_total, n_struct, offsets, X, k, (const int*)ids, flags);
}

}  // namespace pesto
