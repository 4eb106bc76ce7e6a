// pesto_api.hip - C ABI of libpesto_hip.so (declared in include/pesto_hip.h): model handle, grow-only device
// workspace, the forward launch sequence on one HIP stream, per-stage entry points for the parity tests.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "pesto_kernels.h"
#include "pesto_schema.h"

using namespace pesto;

namespace {
thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(PESTO_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// grow-only device buffer
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) { if (hipFree(p) != hipSuccess) return -1; p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 4 + 256;
        if (hipMalloc(&p, want) != hipSuccess) return -1;
        cap = want;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};
}  // namespace

struct pesto_model {
    pesto_config cfg;
    int device = 0;
    hipStream_t stream = nullptr;   // the model's own stream
    DeviceImage img;                // host copy of tables (img.data released after upload)
    float* W = nullptr;             // device weight image
    // workspace (library owns weights + a grow-only workspace; no allocation once warm)
    DevBuf ids_s, geo, q_a, p_a, q_b, p_b, pool_a, seg, z, flags;
    DevBuf rec_nb, rec_cen, zrec;          // MFMA path: per-atom neighbour / centre records and attention sums
    DevBuf rec_nb2;                        // second neighbour-record buffer: an edge launch writes the next layer's records while its own are read
    int precision = PESTO_PRECISION_AUTO;  // pesto_config.precision / pesto_set_precision
    int impl = 2;                          // 2 = MFMA layer (default), 1 = LDS-tiled VALU layer (pesto_debug_select: debug twin)
    int edge_blocks = 512;                 // persistent workgroups of the edge kernel (2 per CU)
    bool knn_brute = false;                // pesto_debug_select: brute-force k-NN for every structure
    int edge_mode = 0;                     // pesto_debug_edge_mode: 0 = per launch, 1 = rendezvous, 2 = node waves
    int64_t n_forward = 0, n_rerun = 0;    // launch sequences run / STRUCTURES repeated on the exact fp32 kernels after a range overflow
    int64_t n_struct_auto = 0;             // structures (frames, collated calls) forwarded on the split kernels under PESTO_PRECISION_AUTO
    DevBuf sflags;                         // range guard: one word per structure (frame) of the launch (SatCtx)
    std::vector<int> h_sflags;             // host copy (counting the structures a repeat covers)
    hipEvent_t ev_flags = nullptr;         // recorded behind the early copy of the flags word (run_forward): AUTO's check does not wait for the pool kernels
    bool early_flags = false;              // that copy was queued by the last run_forward (h_flags[4] holds it once ev_flags has fired)
    std::vector<int> knn_off_host;         // the structure offsets knn_off holds on the device (pesto_knn_collate / pesto_knn_tie_rows)
    bool pad_trigger = true;               // pesto_set_auto_pad_trigger: AUTO repeats structures with zero-padded neighbour slots on the exact kernels (SatCtx::pad_trigger)
    float state_limit = PESTO_AUTO_STATE_LIMIT_DEFAULT;   // pesto_set_auto_state_limit: conditioning trigger of AUTO (SatCtx::state_limit)
    bool async_auto = false;               // pesto_set_async_auto: device-pointer calls under AUTO defer their check to the next call
    // every launch sequence uses the ONE workspace below: sequences on different streams are ordered through this event
    hipEvent_t ws_ev = nullptr;
    hipStream_t ws_stream = nullptr;
    bool ws_pending = false;
    int* h_flags = nullptr;                // pinned host copy of the flags word (read back without a pageable staging copy)
    // PESTO_PRECISION_AUTO with device pointers stays asynchronous: the flags word of the launch is copied to h_flags[8] behind it and
    // looked at by the NEXT call on the handle (resolve_pending): bad inputs are reported there, a range overflow repeats the
    // remembered launch on the exact fp32 kernels (its buffers must still be valid) and makes the handle run fp32 first from then on
    struct Pending { bool active = false; hipStream_t st = nullptr; hipEvent_t ev = nullptr; void* args = nullptr; } pend;   // args: FwdArgs of the launch
    DevBuf col_seg, col_segend;            // pesto_forward_batch: structure of every atom, end offset of every structure
    DevBuf in_X, in_ids, in_q0, in_roa;   // staging for host-pointer calls
    DevBuf in_M, mask_seen;               // pesto_mask_to_segments: host mask staging, one word per residue column
    bool mask_seen_clean = false;         // the buffer has been cleared since it was (re)allocated
    int mask_gen = 0;                     // generation number the valid rows of a call mark their column with (no clearing launch)
    DevBuf knn_off;                       // structure offsets of the last pesto_knn_collate call
    DevBuf knn_grids, knn_cnt, knn_cur, knn_cell, knn_sorted;   // cell grid of the large structures (pesto_knn_collate)
    DevBuf col_meta, col_ids, col_roa;    // pesto_forward_batch: per-structure table, collated ids / residue columns
    DevBuf dmax, roa_f;                   // per-frame max(D) words; residue column per atom of a frame batch
    std::vector<float> pack;              // host packing buffer for strided host frames
    std::vector<uint8_t> qidx_host;       // pesto_forward_batch_submit: block-local feature indices found in a dense one-hot q0
    bool host_only = false;               // pesto_debug_host_only: submit packs and validates but queues no device work
    std::vector<int> seg_host;            // structure end offsets of the last pesto_forward_structures call (H2D source)
    // pesto_forward_batch_submit / _wait: two staging slots (pinned host + device), one copy stream
    struct BatchSlot {
        void* h_in = nullptr; size_t h_cap = 0;         // pinned: [meta | X | ids | q | roa] of the launch, one H2D copy
        float* h_z = nullptr; size_t hz_cap = 0;         // pinned: the logits of the launch
        int* h_flag = nullptr;                           // pinned: the flags word
        int* h_sflags = nullptr; size_t hs_cap = 0;      // pinned: the per-structure range-guard words of the launch
        DevBuf d_in, d_z;
        hipEvent_t ev_h2d = nullptr, ev_done = nullptr;
        bool busy = false;
        int n_struct = 0, ids_kind = 0, n_index = 0, mode = 0;
        int index_offsets[3] = {0, 0, 0};
        int64_t NT = 0, RT = 0;
        size_t off_X = 0, off_ids = 0, off_q = 0, off_roa = 0;
        std::vector<int> roff, rcount;
        std::vector<float*> z_user;
    } slot[2];
    hipStream_t copy_stream = nullptr;
    int next_slot = 0;
    // state left by pesto_stage_unpack for pesto_stage_layer
    int64_t stage_N = -1;
    // timing
    bool timing = false;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    std::vector<hipEvent_t> kev;           // timing level 2: one event between consecutive layer launches
    std::vector<int> kev_class;            // class of the launch that follows kev[i]: 0 = node, 1..4 = edge nn 8/16/32/64
    int timing_level = 1;
    int n_layer_launches = 0;
    bool have_timing = false;
};

namespace {

size_t ws_bytes(int64_t N, int64_t R) {
    const size_t N1 = (size_t)N + 1;
    return N1 * KMAX * 4 + N1 * KMAX * 16 + 2 * (N1 * S * 4 + N1 * 96 * 4) + (size_t)N * 8 * 4 + (size_t)R * 8 + (size_t)R * 32 * 4 + 64 +
           N1 * (REC_NB + REC_A + REC_CEN + REC_Z) * 4;
}

int ensure_workspace(pesto_model* m, int64_t N, int64_t R) {
    const size_t N1 = (size_t)N + 1;
    // the layer kernels address the per-atom arrays with 32-bit byte offsets behind a buffer resource (pesto_mfma_common.h): the largest,
    // the centre records (REC_CEN floats per atom), must stay inside the 4 GB window
    if (m->impl == 2 && N1 * REC_CEN * sizeof(float) > 0xfffffff0ull)
        return fail(PESTO_ERR_INVALID, "N=%lld atoms in one launch exceed the layer kernels' limit of %lld (split the batch)", (long long)N,
                    (long long)(0xfffffff0ull / (REC_CEN * sizeof(float)) - 1));
    int rc = 0;
    rc |= m->ids_s.ensure(N1 * KMAX * sizeof(int));
    rc |= m->geo.ensure(N1 * KMAX * sizeof(float4));
    rc |= m->q_a.ensure(N1 * S * sizeof(float));
    rc |= m->p_a.ensure(N1 * 96 * sizeof(float));
    rc |= m->q_b.ensure(N1 * S * sizeof(float));
    rc |= m->p_b.ensure(N1 * 96 * sizeof(float));
    rc |= m->pool_a.ensure((size_t)N * 8 * sizeof(float));
    rc |= m->seg.ensure((size_t)(R > 0 ? R : 1) * 2 * sizeof(int));
    rc |= m->z.ensure((size_t)(R > 0 ? R : 1) * 32 * sizeof(float));
    rc |= m->flags.ensure(64);
    if (m->impl == 2) {
        rc |= m->rec_nb.ensure(N1 * REC_NB * sizeof(float));
        rc |= m->rec_cen.ensure(N1 * REC_CEN * sizeof(float));
        rc |= m->rec_nb2.ensure(N1 * REC_A * sizeof(float));
        rc |= m->zrec.ensure((size_t)N1 * REC_Z * sizeof(float));
    }
    return rc ? fail(PESTO_ERR_NOMEM, "device workspace allocation failed for N=%lld R=%lld", (long long)N, (long long)R) : 0;
}

// the largest neighbourhood a layer of the model gathers from (<= KMAX)
int max_nn(const pesto_model* m) {
    int v = 0;
    for (int l = 0; l < m->cfg.n_layers; ++l) v = std::max(v, m->cfg.nn[l]);
    return v;
}

// flags buffer layout: [1] error flag (int); per-frame max(D) bit patterns live in m->dmax
unsigned* dmax_ptr(pesto_model* m) { return m->dmax.as<unsigned>(); }
int* err_ptr(pesto_model* m) { return m->flags.as<int>() + 1; }

// Workspace ordering (one workspace per handle, any number of caller streams): a sequence on stream st first waits for the
// event the previous sequence recorded if that one ran on a different stream, and records the event when it has been queued.
int begin_sequence(pesto_model* m, hipStream_t st) {
    if (m->ws_pending && m->ws_stream != st) HIP_TRY(hipStreamWaitEvent(st, m->ws_ev, 0));
    return 0;
}
int end_sequence(pesto_model* m, hipStream_t st) {
    HIP_TRY(hipEventRecord(m->ws_ev, st));
    m->ws_stream = st;
    m->ws_pending = true;
    return 0;
}
struct Sequence {      // every entry point that touches the workspace holds one for its duration
    pesto_model* m;
    hipStream_t st;
    int rc;
    Sequence(pesto_model* m_, hipStream_t st_) : m(m_), st(st_), rc(begin_sequence(m_, st_)) {}
    ~Sequence() { if (rc == 0) (void)end_sequence(m, st); }
    Sequence(const Sequence&) = delete;
    Sequence& operator=(const Sequence&) = delete;
};

// synchronises st; *flag_out (optional) receives the flags word: bit 0 bad ids, bit 1 bad residue column, bit 2 f16-range overflow
// early: the launch sequence just queued copied its (final) flags word in front of its pool kernels (run_forward): wait for that copy only
int check_device_flag(pesto_model* m, hipStream_t st, int* flag_out = nullptr, int ignore = 0, bool early = false) {
    int flag;
    if (early && m->early_flags) {
        HIP_TRY(hipEventSynchronize(m->ev_flags));
        flag = m->h_flags[4];
    } else {
        HIP_TRY(hipMemcpyAsync(m->h_flags, err_ptr(m), sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        flag = *m->h_flags;
    }
    if (flag_out) *flag_out = flag;
    if (flag & 1) return fail(PESTO_ERR_INVALID, "ids_topk contains an index outside [0, N]");
    if (flag & 2) return fail(PESTO_ERR_INVALID, "res_of_atom contains an index outside [0, R) (from pesto_mask_to_segments: a row of M with != 1 member or an empty residue column)");
    if ((flag & 4) && !(ignore & 4))
        return fail(PESTO_ERR_RANGE, "an activation left the f16 range of the split-MFMA path (z is NaN): use PESTO_PRECISION_AUTO or "
                                     "PESTO_PRECISION_FP32");
    return 0;
}

// one launch sequence = F coordinate frames of N atoms / R residues sharing ids, q0 and the residue map (F = 1: the plain
// collated batch of Model.forward). All pointers are device pointers; X strides are in floats.
struct FwdArgs;
int resolve_pending(pesto_model* m);
struct FwdArgs {
    int64_t N = 0, R = 0, F = 1;
    int k = 0;
    const float* X = nullptr;
    int64_t xs_frame = 0, xs_atom = 3;
    const void* ids = nullptr;
    int ids_kind = PESTO_IDS_INT64;
    const float* q0 = nullptr;
    const int* roa = nullptr;
    float* z_out = nullptr;     // [F*R, n_out]
    // ragged structures with separate-call semantics (pesto_forward_batch, PESTO_BATCH_INDEPENDENT; F = 1 only)
    int n_seg = 0;
    const int* seg_of_atom = nullptr;
    const int* seg_end = nullptr;
};

// the launch sequence of Model.forward (model/model.py:32-52) on stream st
// exact: the state-update layers on the exact fp32 MFMA kernels instead of the f16-split ones
// masked (exact only): the fp32 repeat of PESTO_PRECISION_AUTO - the per-structure guard words of the launch being repeated are in place
// (m->sflags, not cleared) and only the logits of the structures whose word is set are written
int run_forward(pesto_model* m, hipStream_t st, const FwdArgs& a, bool exact, bool masked = false) {
    const int64_t NT = a.N * a.F, RT = a.R * a.F;
    const int N1 = (int)NT + 1;
    const int edge_variant = exact ? 1 : 0;
    const size_t n_dmax = a.seg_of_atom ? (size_t)a.n_seg : (size_t)a.F;
    float* q[2] = {m->q_a.as<float>(), m->q_b.as<float>()};
    float* p[2] = {m->p_a.as<float>(), m->p_b.as<float>()};
    // per-forward words cleared by ONE memset: [max(D) bit patterns, n_dmax | segment bounds of the pool layer: lo_enc RT, hi RT]
    const size_t seg_off = (n_dmax + 3) / 4 * 4;      // (in ints)
    const size_t clear_bytes = ((seg_off + 2 * (size_t)RT) * 4 + 63) / 64 * 64;      // (a multiple of 64 bytes: one fill kernel, no tail launch)
    if (m->dmax.ensure(clear_bytes) || m->sflags.ensure(n_dmax * 4)) return fail(PESTO_ERR_NOMEM, "workspace allocation failed");
    // conditioning trigger: only where a flagged structure is repeated (AUTO on the split kernels); f16_split never repeats, so it never flags
    const float state_limit = (!exact && m->precision == PESTO_PRECISION_AUTO && m->state_limit > 0.0f) ? m->state_limit : __builtin_huge_valf();
    const SatCtx sc{err_ptr(m), m->sflags.as<int>(), a.seg_of_atom, (!a.seg_of_atom && a.F > 1) ? (int)a.N : 0, state_limit,
                    (!exact && m->precision == PESTO_PRECISION_AUTO && m->pad_trigger) ? 1 : 0, (exact && masked) ? 1 : 0};
    int* seg_lo = m->dmax.as<int>() + seg_off;
    int* seg_hi = seg_lo + RT;
    const bool bounds_in_embed = a.F == 1;             // found by the unpack launch (trajectory batches expand res_of_atom per frame behind it: separate launches)
    m->n_forward += 1;
    if (!exact && m->precision == PESTO_PRECISION_AUTO) m->n_struct_auto += (int64_t)(a.seg_of_atom ? a.n_seg : a.F);      // (= the launch's guard words)
    const int* roa = a.roa;
    if (a.F > 1) {
        if (m->roa_f.ensure((size_t)NT * 4)) return fail(PESTO_ERR_NOMEM, "workspace allocation failed");
        roa = m->roa_f.as<int>();
    }
    if (m->timing) HIP_TRY(hipEventRecord(m->ev[0], st));
    // no fill launches: the embed launch (first of the forward) clears the flag words and the per-forward words above; the unpack launch
    // behind it finds the pool's segment bounds (40 -> 38 launches per forward with the two changes)
    SegBoundsArgs sb;
    if (bounds_in_embed) sb = SegBoundsArgs{a.roa, seg_lo, seg_hi, (int)RT, err_ptr(m)};
    launch_embed(st, m->W, m->img.model.em, (int)NT, (int)a.N, m->cfg.n0, a.q0, q[0], p[0],      // also p0 = zeros and the sink rows (model.py:37, model_operations.py:17)
                 ClearArgs{m->flags.as<int>(), 2, m->dmax.as<int>(), (int)(clear_bytes / 4), m->sflags.as<int>(), masked ? 0 : (int)n_dmax}, sc);
    // small launches of the shipped path: pass 2 of the geometry rides in the node launch that writes the first layer's records (one
    // dependent launch less per forward: 38 -> 37)
    const bool merge_u2 = !exact && m->impl == 2 && unpack2_merge_blocks((int)NT, N1) > 0;
    launch_unpack(st, (int)a.N, (int)a.F, a.k, a.X, a.xs_frame, a.xs_atom, a.ids, a.ids_kind, m->ids_s.as<int>(), m->geo.as<float4>(),
                  dmax_ptr(m), err_ptr(m), a.seg_of_atom, a.seg_end, sb, merge_u2, max_nn(m));
    if (a.F > 1) launch_expand_roa(st, (int)a.N, (int)a.R, (int)a.F, a.roa, m->roa_f.as<int>(), err_ptr(m));
    if (m->timing) HIP_TRY(hipEventRecord(m->ev[1], st));
    int cur = 0;
    const bool detail = m->timing && m->timing_level >= 2 && m->impl == 2;
    size_t kevi = 0;
    auto mark = [&](int cls) -> hipError_t {      // event in front of the next layer launch (detail timing only)
        if (!detail) return hipSuccess;
        if (kevi == m->kev.size()) {
            hipEvent_t e;
            if (hipError_t rc = hipEventCreate(&e)) return rc;
            m->kev.push_back(e); m->kev_class.push_back(cls);
        }
        m->kev_class[kevi] = cls;
        return hipEventRecord(m->kev[kevi++], st);
    };
    auto nn_class = [](int nn) { return nn == 8 ? 1 : nn == 16 ? 2 : nn == 32 ? 3 : 4; };
    if (m->impl == 2 && edge_variant == 0) {
        // shipped path: ONE node launch (the first layer's records), then one edge launch per layer - edges and attention, the layer's
        // output MLPs (finish phase: new state into the other half of the ping-pong pair) and the NEXT layer's records (prepare phase).
        // Neighbour records (gathered by every workgroup) ping-pong; the centre records are rewritten in place (only the wave that
        // owns a centre reads its record, before the same workgroup iteration writes the next layer's).
        float* rnb[2] = {m->rec_nb.as<float>(), m->rec_nb2.as<float>()};
        float* rcen = m->rec_cen.as<float>();
        const int L = m->cfg.n_layers;
        HIP_TRY(mark(0));
        launch_node(st, m->W, nullptr, &m->img.layers[0], N1, q[0], p[0], m->zrec.as<float>(), rnb[0], rcen, edge_variant, err_ptr(m),
                    merge_u2 ? Unpack2Args{(int)NT, (int)a.N, m->ids_s.as<int>(), m->geo.as<float4>(), dmax_ptr(m), a.seg_of_atom} : Unpack2Args());
        for (int l = 0; l < L; ++l) {
            HIP_TRY(mark(nn_class(m->cfg.nn[l])));
#ifdef PESTO_ABL_NOPREP      // timing-only ablation: no prepare phase (every layer reads the first layer's records)
            launch_edge(st, m->W, m->img.layers[l], N1, m->ids_s.as<int>(), m->geo.as<float4>(), rnb[0], rcen, p[cur], m->zrec.as<float>(), m->edge_blocks,
                        edge_variant, err_ptr(m), q[cur], q[cur ^ 1], p[cur ^ 1], nullptr, rnb[1], rcen,
#else
            launch_edge(st, m->W, m->img.layers[l], N1, m->ids_s.as<int>(), m->geo.as<float4>(), rnb[cur], rcen, p[cur], m->zrec.as<float>(), m->edge_blocks,
                        edge_variant, err_ptr(m), q[cur], q[cur ^ 1], p[cur ^ 1],
                        l + 1 < L ? &m->img.layers[l + 1] : nullptr, rnb[cur ^ 1], rcen,
#endif
                        m->edge_mode);
            cur ^= 1;
        }
        HIP_TRY(mark(-1));
        if (detail) { m->kev.resize(kevi); m->kev_class.resize(kevi); }
    } else if (m->impl == 2) {
        // per layer: node kernel (finish layer l-1, write layer l's records) then edge kernel; state updated in place
        for (int l = 0; l < m->cfg.n_layers; ++l) {
            HIP_TRY(mark(0));
            launch_node(st, m->W, l > 0 ? &m->img.layers[l - 1] : nullptr, &m->img.layers[l], N1, q[0], p[0], m->zrec.as<float>(),
                        m->rec_nb.as<float>(), m->rec_cen.as<float>(), edge_variant, err_ptr(m));
            HIP_TRY(mark(nn_class(m->cfg.nn[l])));
            launch_edge(st, m->W, m->img.layers[l], N1, m->ids_s.as<int>(), m->geo.as<float4>(), m->rec_nb.as<float>(),
                        m->rec_cen.as<float>(), p[0], m->zrec.as<float>(), m->edge_blocks, edge_variant, err_ptr(m));
        }
        HIP_TRY(mark(0));
        launch_node(st, m->W, &m->img.layers[m->cfg.n_layers - 1], nullptr, N1, q[0], p[0], m->zrec.as<float>(), m->rec_nb.as<float>(),
                    m->rec_cen.as<float>(), edge_variant, err_ptr(m));
        HIP_TRY(mark(-1));
        if (detail) { m->kev.resize(kevi); m->kev_class.resize(kevi); }
    } else {
        for (int l = 0; l < m->cfg.n_layers; ++l) {
            launch_layer_v1(st, m->W, m->img.layers[l], N1, m->ids_s.as<int>(), m->geo.as<float4>(), q[cur], p[cur], q[cur ^ 1], p[cur ^ 1]);
            cur ^= 1;
        }
    }
    if (m->timing) {
        HIP_TRY(hipEventRecord(m->ev[2], st));
        m->n_layer_launches = m->impl != 2 ? m->cfg.n_layers : edge_variant == 0 ? m->cfg.n_layers + 1 : 2 * m->cfg.n_layers + 1;
        m->have_timing = true;
    }
    // The flags word is FINAL here: bad ids / residue columns are found by the unpack (and frame-expansion) launches, the range guard fires
    // in the layer kernels; the pool kernels only read it (they are fp32: nothing in them can overflow). AUTO's check of a device-pointer
    // call therefore waits for THIS copy, not for the stream: the host gets its answer while the pool kernels still run and the next
    // call's launches queue up behind them - no idle gap between two one-structure forwards (round 5, batch-1 latency).
    m->early_flags = false;
    if (!exact && !masked && m->precision == PESTO_PRECISION_AUTO) {
        if (!m->ev_flags) HIP_TRY(hipEventCreateWithFlags(&m->ev_flags, hipEventDisableTiming));
        HIP_TRY(hipMemcpyAsync(m->h_flags + 4, err_ptr(m), sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipEventRecord(m->ev_flags, st));
        m->early_flags = true;
    }
    launch_pool(st, m->W, m->img.model, m->cfg.n_out, (int)NT, (int)RT, q[cur] + S, p[cur] + 96, roa, m->pool_a.as<float>(),
                seg_lo, seg_hi, err_ptr(m), nullptr, nullptr, a.z_out, bounds_in_embed, sc, masked);
    HIP_TRY(hipGetLastError());
    return 0;
}

int check_model(const pesto_model* m) { return m ? 0 : fail(PESTO_ERR_INVALID, "null model handle"); }

// number of structures (frames) of a launch = words of its range guard
size_t n_guard_words(const FwdArgs& a) { return a.seg_of_atom ? (size_t)a.n_seg : (size_t)a.F; }

// the fp32 repeat of a launch whose range guard fired (PESTO_PRECISION_AUTO): the exact kernels run the whole launch again - its
// per-structure guard words are still in m->sflags - and the pool kernel writes ONLY the logits of the flagged structures. The others
// keep the logits of the split kernels, so a structure's bits do not depend on what shared its launch. Counts the structures repeated.
int rerun_flagged(pesto_model* m, hipStream_t st, const FwdArgs& a) {
    const size_t n = n_guard_words(a);
    m->h_sflags.assign(n, 0);
    HIP_TRY(hipMemcpyAsync(m->h_sflags.data(), m->sflags.p, n * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (size_t i = 0; i < n; ++i) m->n_rerun += (m->h_sflags[i] & 4) ? 1 : 0;
    return run_forward(m, st, a, true, true);
}

// One forward under the handle's precision policy, then flag handling.
//   FP32      : exact kernels. F16_SPLIT: split kernels. AUTO: split kernels; the structures whose range guard fired are computed
//               again on the exact kernels (rerun_flagged; the inputs are still in place).
//   sync_check: read the flags word back (synchronises st) and turn bad inputs / a range overflow into an error code / the repeat.
//               Host-pointer calls always do (they wait for the D2H copy anyway; `after_run` queues that copy). Device-pointer calls
//               do in AUTO mode - F16_SPLIT and FP32 stay asynchronous and unchecked there (bad inputs make every logit NaN, an
//               overflow the logits of its structure: written by the pool kernel).
//   defer     : AUTO with device pointers on a handle with pesto_set_async_auto(m, 1): no synchronisation, the check is made by the
//               next call on the handle (resolve_pending).
template <typename AfterRun>
int forward_policy(pesto_model* m, hipStream_t st, const FwdArgs& a, bool sync_check, AfterRun after_run, bool defer = false, bool device_call = false) {
    const bool exact_first = m->precision == PESTO_PRECISION_FP32 || m->impl != 2;
    if (int rc = run_forward(m, st, a, exact_first)) return rc;
    if (int rc = after_run()) return rc;
    if (defer && m->precision == PESTO_PRECISION_AUTO && !exact_first) {
        // the flags word travels to pinned memory behind the launch; until the next call on the handle has looked at it, the
        // structures of an overflowed launch hold NaN logits (written by the pool kernel)
        if (!m->pend.ev) HIP_TRY(hipEventCreateWithFlags(&m->pend.ev, hipEventDisableTiming));
        HIP_TRY(hipMemcpyAsync(m->h_flags + 8, err_ptr(m), sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipEventRecord(m->pend.ev, st));
        if (!m->pend.args) m->pend.args = new FwdArgs();
        *static_cast<FwdArgs*>(m->pend.args) = a;
        m->pend.st = st;
        m->pend.active = true;
        return 0;
    }
    if (!sync_check) return 0;
    int flag = 0;
    const bool may_rerun = m->precision == PESTO_PRECISION_AUTO && !exact_first;
    // (device-pointer calls have nothing to wait for but the verdict: the early copy; host-pointer calls wait for their D2H anyway)
    if (int rc = check_device_flag(m, st, &flag, may_rerun ? 4 : 0, device_call && may_rerun)) return rc;
    if ((flag & 4) && may_rerun) {
        if (int rc = rerun_flagged(m, st, a)) return rc;
        if (int rc = after_run()) return rc;
        return check_device_flag(m, st);
    }
    return 0;
}

// the deferred check of the last asynchronous AUTO launch (see forward_policy): called at the start of every entry point of the handle
int resolve_pending(pesto_model* m) {
    if (!m->pend.active) return 0;
    m->pend.active = false;
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipEventSynchronize(m->pend.ev));
    const int flag = m->h_flags[8];
    if (flag & 1) return fail(PESTO_ERR_INVALID, "the previous (asynchronous) forward had an index of ids_topk outside [0, N]: its logits are NaN");
    if (flag & 2) return fail(PESTO_ERR_INVALID, "the previous (asynchronous) forward had an index of res_of_atom outside [0, R) (from pesto_mask_to_segments: "
                                               "a row of M with != 1 member or an empty residue column): its logits are NaN");
    if (flag & 4) {
        // repeat the flagged structures on the exact kernels, on the launch's own stream, into the same z (the caller keeps the buffers
        // of an asynchronous call valid until the next call on the handle or pesto_synchronize returns; the launch's guard words and
        // structure table are still in the workspace: nothing has been queued on the handle since)
        Sequence seq(m, m->pend.st);
        if (seq.rc) return seq.rc;
        if (int rc = rerun_flagged(m, m->pend.st, *static_cast<FwdArgs*>(m->pend.args))) return rc;
    }
    return 0;
}

}  // namespace

extern "C" {

const char* pesto_last_error(void) { return g_err.c_str(); }

int pesto_blob_size(const pesto_config* cfg, int64_t* n_floats) {
    if (!config_ok(cfg) || !n_floats) return fail(PESTO_ERR_INVALID, "invalid pesto_config");
    *n_floats = host_schema(*cfg).total;
    return 0;
}

int pesto_create(const pesto_config* cfg, const float* weights, int64_t n_weights, int device, pesto_model** out) {
    if (!out) return fail(PESTO_ERR_INVALID, "out is null");
    *out = nullptr;
    if (!config_ok(cfg)) return fail(PESTO_ERR_INVALID, "invalid pesto_config (nn must be 8/16/32/64, depths 1 or 3, n_out <= 32, precision a pesto_precision)");
    const int64_t need = host_schema(*cfg).total;
    if (!weights || n_weights != need)
        return fail(PESTO_ERR_INVALID, "weight blob has %lld floats, config needs %lld", (long long)n_weights, (long long)need);
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(PESTO_ERR_INVALID, "device %d out of range (%d visible)", device, ndev);
    HIP_TRY(hipSetDevice(device));
    pesto_model* m = new pesto_model();
    m->cfg = *cfg;
    m->device = device;
    m->img = build_device_image(*cfg, weights);
    m->precision = cfg->precision;
    hipError_t e = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&m->ws_ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipHostMalloc((void**)&m->h_flags, 64, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc((void**)&m->W, m->img.data.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(m->W, m->img.data.data(), m->img.data.size() * sizeof(float), hipMemcpyHostToDevice);
    for (int i = 0; i < 3 && e == hipSuccess; ++i) e = hipEventCreate(&m->ev[i]);
    if (e != hipSuccess) {
        pesto_destroy(m);
        return fail(PESTO_ERR_HIP, "device setup failed: %s", hipGetErrorString(e));
    }
    std::vector<float>().swap(m->img.data);
    *out = m;
    return 0;
}

int pesto_destroy(pesto_model* m) {
    if (!m) return 0;
    (void)hipSetDevice(m->device);
    if (m->stream) { (void)hipStreamSynchronize(m->stream); (void)hipStreamDestroy(m->stream); }
    (void)hipDeviceSynchronize();
    for (auto& b : m->slot) {
        if (b.ev_h2d) (void)hipEventDestroy(b.ev_h2d);
        if (b.ev_done) (void)hipEventDestroy(b.ev_done);
        if (b.h_in) (void)hipHostFree(b.h_in);
        if (b.h_z) (void)hipHostFree(b.h_z);
        if (b.h_flag) (void)hipHostFree(b.h_flag);
        if (b.h_sflags) (void)hipHostFree(b.h_sflags);
        b.d_in.release(); b.d_z.release();
    }
    if (m->copy_stream) (void)hipStreamDestroy(m->copy_stream);
    if (m->pend.ev) (void)hipEventDestroy(m->pend.ev);
    delete static_cast<FwdArgs*>(m->pend.args);
    if (m->ws_ev) (void)hipEventDestroy(m->ws_ev);
    if (m->ev_flags) (void)hipEventDestroy(m->ev_flags);
    if (m->h_flags) (void)hipHostFree(m->h_flags);
    for (auto& e : m->ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : m->kev) if (e) (void)hipEventDestroy(e);
    if (m->W) (void)hipFree(m->W);
    for (DevBuf* b : {&m->ids_s, &m->geo, &m->q_a, &m->p_a, &m->q_b, &m->p_b, &m->pool_a, &m->seg, &m->z, &m->flags,
                      &m->in_X, &m->in_ids, &m->in_q0, &m->in_roa, &m->rec_nb, &m->rec_cen, &m->rec_nb2, &m->zrec, &m->knn_off, &m->dmax, &m->roa_f, &m->col_meta, &m->col_ids, &m->col_roa, &m->knn_grids, &m->knn_cnt, &m->knn_cur, &m->knn_cell, &m->knn_sorted, &m->col_seg, &m->col_segend, &m->in_M, &m->mask_seen, &m->sflags})
        b->release();
    delete m;
    return 0;
}

int pesto_workspace_bytes(const pesto_model* m, int64_t N, int64_t R, int64_t* bytes) {
    if (check_model(m)) return PESTO_ERR_INVALID;
    if (N < 1 || R < 1 || !bytes) return fail(PESTO_ERR_INVALID, "bad arguments");
    *bytes = (int64_t)ws_bytes(N, R);
    return 0;
}

int pesto_synchronize(pesto_model* m) {
    if (check_model(m)) return PESTO_ERR_INVALID;
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t pst = m->pend.active ? m->pend.st : nullptr;
    const bool had = m->pend.active;
    const int rc = resolve_pending(m);          // deferred check of the last asynchronous AUTO launch (may queue its fp32 repeat)
    if (had) HIP_TRY(hipStreamSynchronize(pst));
    HIP_TRY(hipStreamSynchronize(m->stream));
    return rc;
}

int pesto_set_precision(pesto_model* m, int32_t precision) {
    if (check_model(m)) return PESTO_ERR_INVALID;
    if (precision != PESTO_PRECISION_AUTO && precision != PESTO_PRECISION_F16_SPLIT && precision != PESTO_PRECISION_FP32)
        return fail(PESTO_ERR_INVALID, "precision must be PESTO_PRECISION_AUTO, _F16_SPLIT or _FP32");
    if (int rc = resolve_pending(m)) return rc;
    m->precision = precision;
    return 0;
}

int pesto_set_async_auto(pesto_model* m, int32_t enabled) {
    if (check_model(m)) return PESTO_ERR_INVALID;
    if (int rc = resolve_pending(m)) return rc;
    m->async_auto = enabled != 0;
    return 0;
}

int pesto_set_auto_state_limit(pesto_model* m, float limit) {
    if (check_model(m)) return PESTO_ERR_INVALID;
    if (limit != limit) return fail(PESTO_ERR_INVALID, "state limit must be a number (<= 0 switches the trigger off)");
    if (int rc = resolve_pending(m)) return rc;
    m->state_limit = limit;
    return 0;
}

int pesto_set_auto_pad_trigger(pesto_model* m, int32_t enabled) {
    if (check_model(m)) return PESTO_ERR_INVALID;
    if (int rc = resolve_pending(m)) return rc;
    m->pad_trigger = enabled != 0;
    return 0;
}

int pesto_get_status(const pesto_model* m, int32_t* precision, int64_t* n_forward, int64_t* n_fp32_rerun) {
    if (check_model(m)) return PESTO_ERR_INVALID;
    if (int rc = resolve_pending(const_cast<pesto_model*>(m))) return rc;
    if (precision) *precision = m->precision;
    if (n_forward) *n_forward = m->n_forward;
    if (n_fp32_rerun) *n_fp32_rerun = m->n_rerun;
    return 0;
}

int pesto_get_auto_counters(const pesto_model* m, int64_t* n_structures, int64_t* n_repeated) {
    if (check_model(m)) return PESTO_ERR_INVALID;
    if (int rc = resolve_pending(const_cast<pesto_model*>(m))) return rc;
    if (n_structures) *n_structures = m->n_struct_auto;
    if (n_repeated) *n_repeated = m->n_rerun;
    return 0;
}

int pesto_debug_select(pesto_model* m, int32_t layer_kernels, int32_t knn_brute_force) {
    if (check_model(m)) return PESTO_ERR_INVALID;
    if (layer_kernels != 0 && layer_kernels != 1) return fail(PESTO_ERR_INVALID, "layer_kernels must be 0 or 1");
    m->impl = layer_kernels == 1 ? 1 : 2;
    m->knn_brute = knn_brute_force != 0;
    return 0;
}

int pesto_debug_host_only(pesto_model* m, int32_t enabled) {
    if (check_model(m)) return PESTO_ERR_INVALID;
    m->host_only = enabled != 0;
    return 0;
}

int pesto_debug_edge_mode(pesto_model* m, int32_t mode) {
    if (check_model(m)) return PESTO_ERR_INVALID;
    if (mode < 0 || mode > 2) return fail(PESTO_ERR_INVALID, "mode must be 0 (per launch), 1 (rendezvous) or 2 (node waves)");
    m->edge_mode = mode;
    return 0;
}

int pesto_set_timing(pesto_model* m, int32_t enabled) {
    if (check_model(m)) return PESTO_ERR_INVALID;
    m->timing = enabled != 0;
    m->timing_level = enabled >= 2 ? 2 : 1;
    m->have_timing = false;
    return 0;
}

int pesto_get_timing(pesto_model* m, double* layers_ms, double* total_ms, int32_t* n_layer_launches) {
    if (check_model(m)) return PESTO_ERR_INVALID;
    if (!m->have_timing) return fail(PESTO_ERR_STATE, "no timed forward has run (pesto_set_timing + pesto_forward)");
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipEventSynchronize(m->ev[2]));
    float a = 0.f, b = 0.f;
    HIP_TRY(hipEventElapsedTime(&a, m->ev[1], m->ev[2]));
    HIP_TRY(hipEventElapsedTime(&b, m->ev[0], m->ev[2]));
    if (layers_ms) *layers_ms = a;
    if (total_ms) *total_ms = b;
    if (n_layer_launches) *n_layer_launches = m->n_layer_launches;
    return 0;
}

int pesto_get_kernel_timing(pesto_model* m, double ms_sum[5], int32_t launches[5]) {
    if (check_model(m)) return PESTO_ERR_INVALID;
    if (!m->have_timing || m->timing_level < 2 || m->kev.size() < 2) return fail(PESTO_ERR_STATE, "no forward timed at level 2 (pesto_set_timing(m, 2))");
    if (!ms_sum || !launches) return fail(PESTO_ERR_INVALID, "bad arguments");
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipEventSynchronize(m->kev.back()));
    for (int c = 0; c < 5; ++c) { ms_sum[c] = 0.0; launches[c] = 0; }
    for (size_t i = 0; i + 1 < m->kev.size(); ++i) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, m->kev[i], m->kev[i + 1]));
        const int c = m->kev_class[i];
        if (c >= 0 && c < 5) { ms_sum[c] += ms; launches[c] += 1; }
    }
    return 0;
}

}  // extern "C"

namespace {
// pesto_forward / pesto_forward_frames / pesto_forward_structures: n_struct > 0 (one frame only) gives the structures
// [struct_offsets[s], struct_offsets[s+1]) of the collated batch separate-call semantics (per-structure wrap target and max(D))
int forward_common(pesto_model* m, int64_t N, int64_t R, int32_t k, int64_t n_frames, const float* X, int64_t x_frame_stride,
                   int64_t x_atom_stride, const void* ids_topk, int32_t ids_kind, const float* q0, const int32_t* res_of_atom,
                   float* z_out, int32_t frames_per_launch, int32_t ptr_kind, void* stream, int32_t n_struct, const int32_t* struct_offsets) {
    if (check_model(m)) return PESTO_ERR_INVALID;
    if (int rc = resolve_pending(m)) return rc;
    if (N < 1 || R < 1 || N > 0x7ffffff0 / 96 || R > N) return fail(PESTO_ERR_INVALID, "bad sizes N=%lld R=%lld", (long long)N, (long long)R);
    if (n_frames < 1) return fail(PESTO_ERR_INVALID, "n_frames=%lld must be >= 1", (long long)n_frames);
    if (k < 1 || k > KMAX) return fail(PESTO_ERR_INVALID, "k=%d must be in 1..%d", k, KMAX);
    for (int l = 0; l < m->cfg.n_layers; ++l)
        if (m->cfg.nn[l] > KMAX) return fail(PESTO_ERR_INVALID, "layer nn exceeds %d", KMAX);
    if (ids_kind != PESTO_IDS_INT32 && ids_kind != PESTO_IDS_INT64) return fail(PESTO_ERR_INVALID, "ids_kind must be 32 or 64");
    if (!X || !ids_topk || !q0 || !res_of_atom || !z_out) return fail(PESTO_ERR_INVALID, "null buffer");
    if (ptr_kind != PESTO_PTR_HOST && ptr_kind != PESTO_PTR_DEVICE) return fail(PESTO_ERR_INVALID, "ptr_kind must be PESTO_PTR_HOST or PESTO_PTR_DEVICE");
    if (x_atom_stride < 3 && N > 1) return fail(PESTO_ERR_INVALID, "x_atom_stride must be >= 3 floats");
    // frames per launch: ~24k atoms fill the chip (measured: 8 x 3,000 beats 16 x 3,000 - at 5.6 KB of records and state
    // per atom, 24k atoms = 134 MB still sit in the 256 MB Infinity Cache); chunks are balanced over the frame count
    int64_t fpl = frames_per_launch > 0 ? frames_per_launch : (24576 + N / 2) / N;
    if (fpl < 1) fpl = 1;
    if (fpl > n_frames) fpl = n_frames;
    if (fpl * N > 0x7ffffff0 / 96) fpl = (0x7ffffff0 / 96) / N;
    const int64_t n_chunks = (n_frames + fpl - 1) / fpl;
    fpl = (n_frames + n_chunks - 1) / n_chunks;
    HIP_TRY(hipSetDevice(m->device));
    if (int rc = ensure_workspace(m, N * fpl, R * fpl)) return rc;
    const size_t id_sz = ids_kind == PESTO_IDS_INT64 ? 8 : 4;
    const int n_out = m->cfg.n_out;
    FwdArgs a;
    a.N = N; a.R = R; a.k = k; a.ids_kind = ids_kind;
    if (n_struct > 0) {
        if (n_frames != 1 || !struct_offsets || struct_offsets[0] != 0 || struct_offsets[n_struct] != N)
            return fail(PESTO_ERR_INVALID, "struct_offsets must span [0, N] (one frame)");
        for (int s = 0; s < n_struct; ++s)
            if (struct_offsets[s + 1] <= struct_offsets[s]) return fail(PESTO_ERR_INVALID, "empty or unordered structure %d", s);
        if (m->col_seg.ensure((size_t)N * 4) || m->col_segend.ensure((size_t)n_struct * 4)) return fail(PESTO_ERR_NOMEM, "workspace allocation failed");
    }
    // the structure table is uploaded inside the sequence (it lives in the shared workspace)
    auto upload_segments = [&](hipStream_t st) -> int {
        if (n_struct <= 0) return 0;
        m->seg_host.assign(struct_offsets + 1, struct_offsets + n_struct + 1);
        HIP_TRY(hipMemcpyAsync(m->col_segend.p, m->seg_host.data(), (size_t)n_struct * 4, hipMemcpyHostToDevice, st));
        launch_segments(st, (int)N, n_struct, m->col_segend.as<int>(), m->col_seg.as<int>());
        a.n_seg = n_struct; a.seg_of_atom = m->col_seg.as<int>(); a.seg_end = m->col_segend.as<int>();
        return 0;
    };
    if (ptr_kind == PESTO_PTR_DEVICE) {   // stream is taken literally: NULL is HIP's default (null) stream
        a.ids = ids_topk; a.q0 = q0; a.roa = res_of_atom; a.xs_frame = x_frame_stride; a.xs_atom = x_atom_stride;
        hipStream_t st = (hipStream_t)stream;
        Sequence seq(m, st);
        if (seq.rc) return seq.rc;
        if (int rc = upload_segments(st)) return rc;
        for (int64_t c = 0; c < n_chunks; ++c) {
            const int64_t f0 = c * n_frames / n_chunks;
            a.F = (c + 1) * n_frames / n_chunks - f0;
            a.X = X + f0 * x_frame_stride;
            a.z_out = z_out + f0 * R * n_out;
            // AUTO: checked before the call returns (one 4-byte D2H + a stream synchronisation); on a handle with pesto_set_async_auto the
            // last chunk is left to the next call on the handle instead (every other chunk of a multi-chunk frame call is checked here)
            const bool last = c + 1 == n_chunks;
            if (int rc = forward_policy(m, st, a, m->precision == PESTO_PRECISION_AUTO, [] { return 0; }, last && m->async_auto, true)) return rc;
        }
        return 0;
    }
    hipStream_t st = stream ? (hipStream_t)stream : m->stream;
    if (m->in_X.ensure((size_t)N * fpl * 3 * 4) || m->in_ids.ensure((size_t)N * k * id_sz) || m->in_q0.ensure((size_t)N * m->cfg.n0 * 4) ||
        m->in_roa.ensure((size_t)N * 4))
        return fail(PESTO_ERR_NOMEM, "staging allocation failed");
    Sequence seq(m, st);
    if (seq.rc) return seq.rc;
    if (int rc = upload_segments(st)) return rc;
    HIP_TRY(hipMemcpyAsync(m->in_ids.p, ids_topk, (size_t)N * k * id_sz, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(m->in_q0.p, q0, (size_t)N * m->cfg.n0 * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(m->in_roa.p, res_of_atom, (size_t)N * 4, hipMemcpyHostToDevice, st));
    a.ids = m->in_ids.p; a.q0 = m->in_q0.as<float>(); a.roa = m->in_roa.as<int>();
    a.X = m->in_X.as<float>(); a.xs_frame = 3 * N; a.xs_atom = 3;
    a.z_out = m->z.as<float>();
    const bool packed = x_atom_stride == 3 && (x_frame_stride == 3 * N || n_frames == 1);
    for (int64_t c = 0; c < n_chunks; ++c) {
        const int64_t f0 = c * n_frames / n_chunks;
        a.F = (c + 1) * n_frames / n_chunks - f0;
        const float* src = X + f0 * x_frame_stride;
        if (!packed) {   // gather the strided host frames (e.g. the reference's [N, frames, 3] trajectory tensor) into [F, N, 3]
            m->pack.resize((size_t)a.F * N * 3);
            for (int64_t f = 0; f < a.F; ++f)
                for (int64_t i = 0; i < N; ++i) {
                    const float* x = src + f * x_frame_stride + i * x_atom_stride;
                    float* d = m->pack.data() + ((size_t)f * N + i) * 3;
                    d[0] = x[0]; d[1] = x[1]; d[2] = x[2];
                }
            src = m->pack.data();
        }
        HIP_TRY(hipMemcpyAsync(m->in_X.p, src, (size_t)a.F * N * 3 * 4, hipMemcpyHostToDevice, st));
        float* dst = z_out + f0 * R * n_out;
        const size_t zbytes = (size_t)a.F * R * n_out * 4;
        auto copy_back = [&]() -> int { HIP_TRY(hipMemcpyAsync(dst, m->z.p, zbytes, hipMemcpyDeviceToHost, st)); return 0; };
        if (int rc = forward_policy(m, st, a, true, copy_back)) return rc;   // synchronises: staging buffers are free for the next chunk
    }
    return 0;
}
}  // namespace

extern "C" {

int pesto_forward(pesto_model* m, int64_t N, int64_t R, int32_t k, const float* X, const void* ids_topk, int32_t ids_kind,
                  const float* q0, const int32_t* res_of_atom, float* z_out, int32_t ptr_kind, void* stream) {
    return forward_common(m, N, R, k, 1, X, 3 * N, 3, ids_topk, ids_kind, q0, res_of_atom, z_out, 1, ptr_kind, stream, 0, nullptr);
}

int pesto_forward_structures(pesto_model* m, int64_t N, int64_t R, int32_t k, int32_t n_struct, const int32_t* struct_offsets, const float* X,
                             const void* ids_topk, int32_t ids_kind, const float* q0, const int32_t* res_of_atom, float* z_out,
                             int32_t ptr_kind, void* stream) {
    if (n_struct < 1 || !struct_offsets) return fail(PESTO_ERR_INVALID, "n_struct must be >= 1 and struct_offsets non-null");
    return forward_common(m, N, R, k, 1, X, 3 * N, 3, ids_topk, ids_kind, q0, res_of_atom, z_out, 1, ptr_kind, stream, n_struct, struct_offsets);
}

int pesto_forward_frames(pesto_model* m, int64_t N, int64_t R, int32_t k, int64_t n_frames, const float* X, int64_t x_frame_stride,
                         int64_t x_atom_stride, const void* ids_topk, int32_t ids_kind, const float* q0, const int32_t* res_of_atom,
                         float* z_out, int32_t frames_per_launch, int32_t ptr_kind, void* stream) {
    return forward_common(m, N, R, k, n_frames, X, x_frame_stride, x_atom_stride, ids_topk, ids_kind, q0, res_of_atom, z_out,
                          frames_per_launch, ptr_kind, stream, 0, nullptr);
}

int pesto_forward_batch(pesto_model* m, int32_t n_struct, const int64_t* N, const int64_t* R, const int32_t* k, const float* const* X,
                        const void* const* ids_topk0, int32_t ids_kind, const float* const* q0, const int32_t* const* res_of_atom,
                        float* const* z_out, int32_t batch_mode, void* stream) {
    if (check_model(m)) return PESTO_ERR_INVALID;
    if (int rc = resolve_pending(m)) return rc;
    if (batch_mode != PESTO_BATCH_COLLATED && batch_mode != PESTO_BATCH_INDEPENDENT)
        return fail(PESTO_ERR_INVALID, "batch_mode must be PESTO_BATCH_COLLATED or PESTO_BATCH_INDEPENDENT");
    if (n_struct < 1 || !N || !R || !k || !X || !ids_topk0 || !q0 || !res_of_atom || !z_out) return fail(PESTO_ERR_INVALID, "bad arguments");
    if (ids_kind != PESTO_IDS_INT32 && ids_kind != PESTO_IDS_INT64) return fail(PESTO_ERR_INVALID, "ids_kind must be 32 or 64");
    struct Meta { int off, roff, n, r, k; long long idoff; };
    std::vector<Meta> meta((size_t)n_struct);
    int64_t NT = 0, RT = 0, IT = 0;
    for (int b = 0; b < n_struct; ++b) {
        if (N[b] < 1 || R[b] < 1 || R[b] > N[b] || k[b] < 1 || k[b] > KMAX || !X[b] || !ids_topk0[b] || !q0[b] || !res_of_atom[b] || !z_out[b])
            return fail(PESTO_ERR_INVALID, "structure %d: bad sizes or null buffer (N=%lld R=%lld k=%d)", b, (long long)N[b], (long long)R[b], k[b]);
        meta[b] = Meta{(int)NT, (int)RT, (int)N[b], (int)R[b], k[b], (long long)IT};
        NT += N[b]; RT += R[b]; IT += N[b] * k[b];
        if (NT > 0x7ffffff0 / 96) return fail(PESTO_ERR_INVALID, "batch too large");
    }
    HIP_TRY(hipSetDevice(m->device));
    if (int rc = ensure_workspace(m, NT, RT)) return rc;
    const size_t id_sz = ids_kind == PESTO_IDS_INT64 ? 8 : 4;
    const int n0 = m->cfg.n0, n_out = m->cfg.n_out;
    if (m->in_X.ensure((size_t)NT * 12) || m->in_ids.ensure((size_t)IT * id_sz) || m->in_q0.ensure((size_t)NT * n0 * 4) ||
        m->in_roa.ensure((size_t)NT * 4) || m->col_meta.ensure(meta.size() * sizeof(Meta)) || m->col_ids.ensure((size_t)NT * KMAX * 4) ||
        m->col_roa.ensure((size_t)NT * 4) || m->col_seg.ensure((size_t)NT * 4) || m->col_segend.ensure((size_t)n_struct * 4))
        return fail(PESTO_ERR_NOMEM, "staging allocation failed");
    hipStream_t st = stream ? (hipStream_t)stream : m->stream;
    Sequence seq(m, st);
    if (seq.rc) return seq.rc;
    HIP_TRY(hipMemsetAsync(m->flags.p, 0, 8, st));
    HIP_TRY(hipMemcpyAsync(m->col_meta.p, meta.data(), meta.size() * sizeof(Meta), hipMemcpyHostToDevice, st));
    for (int b = 0; b < n_struct; ++b) {       // per-structure arrays land back to back: the concatenations of dataset.py:93-94
        const Meta& mb = meta[b];
        HIP_TRY(hipMemcpyAsync(m->in_X.as<float>() + (size_t)mb.off * 3, X[b], (size_t)mb.n * 12, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync((char*)m->in_ids.p + (size_t)mb.idoff * id_sz, ids_topk0[b], (size_t)mb.n * mb.k * id_sz, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(m->in_q0.as<float>() + (size_t)mb.off * n0, q0[b], (size_t)mb.n * n0 * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(m->in_roa.as<int>() + mb.off, res_of_atom[b], (size_t)mb.n * 4, hipMemcpyHostToDevice, st));
    }
    launch_collate(st, (int)NT, n_struct, m->col_meta.p, m->in_ids.p, ids_kind, m->in_roa.as<int>(), m->col_ids.as<int>(), m->col_roa.as<int>(),
                   m->col_seg.as<int>(), m->col_segend.as<int>(), err_ptr(m));
    HIP_TRY(hipGetLastError());
    if (int rc = check_device_flag(m, st)) return rc;      // bad ids / residue columns are reported before the forward runs
    FwdArgs a;
    a.N = NT; a.R = RT; a.F = 1; a.k = KMAX; a.X = m->in_X.as<float>(); a.xs_frame = 3 * NT; a.xs_atom = 3;
    a.ids = m->col_ids.p; a.ids_kind = PESTO_IDS_INT32; a.q0 = m->in_q0.as<float>(); a.roa = m->col_roa.as<int>(); a.z_out = m->z.as<float>();
    if (batch_mode == PESTO_BATCH_INDEPENDENT) { a.n_seg = n_struct; a.seg_of_atom = m->col_seg.as<int>(); a.seg_end = m->col_segend.as<int>(); }
    auto copy_back = [&]() -> int {
        for (int b = 0; b < n_struct; ++b)
            HIP_TRY(hipMemcpyAsync(z_out[b], m->z.as<float>() + (size_t)meta[b].roff * n_out, (size_t)meta[b].r * n_out * 4, hipMemcpyDeviceToHost, st));
        return 0;
    };
    return forward_policy(m, st, a, true, copy_back);   // synchronises
}

namespace {
struct CollMeta { int off, roff, n, r, k; long long idoff; };
size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

// the device half of one staged launch on the compute stream: (one-hot expansion,) collate, forward, logits + flags to pinned memory
int queue_slot(pesto_model* m, pesto_model::BatchSlot& b, hipStream_t st, bool exact, bool masked = false) {
    Sequence seq(m, st);
    if (seq.rc) return seq.rc;
    const size_t n_words = b.mode == PESTO_BATCH_INDEPENDENT ? (size_t)b.n_struct : 1;
    if (m->sflags.ensure(n_words * 4)) return fail(PESTO_ERR_NOMEM, "workspace allocation failed");
    // masked repeat: the guard words of THIS launch back into the workspace (another slot's launch has used it in between)
    if (masked) HIP_TRY(hipMemcpyAsync(m->sflags.p, b.h_sflags, n_words * 4, hipMemcpyHostToDevice, st));
    char* base = (char*)b.d_in.p;
    HIP_TRY(hipMemsetAsync(m->flags.p, 0, 8, st));
    const float* q0 = (const float*)(base + b.off_q);
    if (b.n_index > 0) {
        launch_onehot(st, (int)b.NT, m->cfg.n0, b.n_index, (const unsigned char*)(base + b.off_q), b.index_offsets, m->in_q0.as<float>(), err_ptr(m));
        q0 = m->in_q0.as<float>();
    }
    launch_collate(st, (int)b.NT, b.n_struct, base, base + b.off_ids, b.ids_kind, (const int*)(base + b.off_roa), m->col_ids.as<int>(),
                   m->col_roa.as<int>(), m->col_seg.as<int>(), m->col_segend.as<int>(), err_ptr(m));
    FwdArgs a;
    a.N = b.NT; a.R = b.RT; a.F = 1; a.k = KMAX; a.X = (const float*)(base + b.off_X); a.xs_frame = 3 * b.NT; a.xs_atom = 3;
    a.ids = m->col_ids.p; a.ids_kind = PESTO_IDS_INT32; a.q0 = q0; a.roa = m->col_roa.as<int>(); a.z_out = b.d_z.as<float>();
    if (b.mode == PESTO_BATCH_INDEPENDENT) { a.n_seg = b.n_struct; a.seg_of_atom = m->col_seg.as<int>(); a.seg_end = m->col_segend.as<int>(); }
    // run_forward clears the flags word itself: the collate / one-hot checks are repeated by the forward's own validation of ids and
    // residue columns (a bad index reaches it as index 0 + the flag, which the memset would lose) - keep the flag by OR-ing it back
    HIP_TRY(hipMemcpyAsync(m->flags.as<int>() + 2, err_ptr(m), sizeof(int), hipMemcpyDeviceToDevice, st));
    if (int rc = run_forward(m, st, a, exact, masked)) return rc;
    HIP_TRY(hipMemcpyAsync(b.h_z, b.d_z.p, (size_t)b.RT * m->cfg.n_out * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(b.h_flag, m->flags.as<int>() + 1, 2 * sizeof(int), hipMemcpyDeviceToHost, st));     // [forward's flags, collate's flags]
    if (!masked) HIP_TRY(hipMemcpyAsync(b.h_sflags, m->sflags.p, n_words * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipEventRecord(b.ev_done, st));
    return 0;
}
}  // namespace

int pesto_forward_batch_submit(pesto_model* m, int32_t n_struct, const int64_t* N, const int64_t* R, const int32_t* k, const float* const* X,
                               const void* const* ids_topk0, int32_t ids_kind, const float* const* q0, const uint8_t* const* q_index,
                               int32_t n_index, const int32_t* index_offsets, const int32_t* const* res_of_atom, float* const* z_out,
                               int32_t batch_mode, int32_t* ticket) {
    if (check_model(m)) return PESTO_ERR_INVALID;
    if (int rc = resolve_pending(m)) return rc;
    if (batch_mode != PESTO_BATCH_COLLATED && batch_mode != PESTO_BATCH_INDEPENDENT)
        return fail(PESTO_ERR_INVALID, "batch_mode must be PESTO_BATCH_COLLATED or PESTO_BATCH_INDEPENDENT");
    if (n_struct < 1 || !N || !R || !k || !X || !ids_topk0 || (!q0 && !q_index) || !res_of_atom || !z_out || !ticket) return fail(PESTO_ERR_INVALID, "bad arguments");
    // PESTO_IDS_NARROW: int32 / int64 tables (what extract_topology hands out, src/data_encoding.py:98-102) travel as uint16 - narrowed and
    // range-checked by the packer's own pass over them, not by three numpy passes in the caller - when every structure has <= 65,536 atoms
    const bool want_narrow = (ids_kind & PESTO_IDS_NARROW) != 0;
    const int src_kind = ids_kind & ~PESTO_IDS_NARROW;
    if (src_kind != PESTO_IDS_INT32 && src_kind != PESTO_IDS_INT64 && src_kind != PESTO_IDS_UINT16) return fail(PESTO_ERR_INVALID, "ids_kind must be 16, 32 or 64");
    if (want_narrow && src_kind == PESTO_IDS_UINT16) return fail(PESTO_ERR_INVALID, "PESTO_IDS_NARROW goes with int32 / int64 tables");
    bool narrow = want_narrow;
    for (int s_ = 0; s_ < n_struct && narrow; ++s_) narrow = N && N[s_] <= 65536;
    ids_kind = narrow ? PESTO_IDS_UINT16 : src_kind;      // (the kind the staged table has)
    if (q_index && (n_index < 1 || n_index > 3 || !index_offsets)) return fail(PESTO_ERR_INVALID, "q_index needs 1..3 index columns and their block offsets");
    const bool detect = !q_index && n_index > 0;      // dense q0 + the block offsets: find the indices while packing (below)
    if (detect && (n_index > 3 || !index_offsets)) return fail(PESTO_ERR_INVALID, "index_offsets: 1..3 block offsets");
    const int n0 = m->cfg.n0, n_out = m->cfg.n_out;
    if (q_index || detect)
        for (int c = 0; c < n_index; ++c)
            if (index_offsets[c] < 0 || index_offsets[c] >= n0 || (c && index_offsets[c] <= index_offsets[c - 1])) return fail(PESTO_ERR_INVALID, "index_offsets must ascend inside [0, n0)");
    pesto_model::BatchSlot& b = m->slot[m->next_slot];
    if (b.busy) return fail(PESTO_ERR_STATE, "both staging slots are in flight: pesto_forward_batch_wait first");
    const size_t id_sz = ids_kind == PESTO_IDS_INT64 ? 8 : ids_kind == PESTO_IDS_INT32 ? 4 : 2;
    std::vector<CollMeta> meta((size_t)n_struct);
    int64_t NT = 0, RT = 0, IT = 0;
    b.roff.resize(n_struct); b.rcount.resize(n_struct); b.z_user.assign(z_out, z_out + n_struct);
    for (int s_ = 0; s_ < n_struct; ++s_) {
        if (N[s_] < 1 || R[s_] < 1 || R[s_] > N[s_] || k[s_] < 1 || k[s_] > KMAX || !X[s_] || !ids_topk0[s_] || (q_index ? !q_index[s_] : !q0[s_]) ||
            !res_of_atom[s_] || !z_out[s_] || (ids_kind == PESTO_IDS_UINT16 && N[s_] > 65536))
            return fail(PESTO_ERR_INVALID, "structure %d: bad sizes or null buffer (N=%lld R=%lld k=%d)", s_, (long long)N[s_], (long long)R[s_], k[s_]);
        meta[s_] = CollMeta{(int)NT, (int)RT, (int)N[s_], (int)R[s_], k[s_], (long long)IT};
        b.roff[s_] = (int)RT; b.rcount[s_] = (int)R[s_];
        NT += N[s_]; RT += R[s_]; IT += N[s_] * k[s_];
        if (NT > 0x7ffffff0 / 96) return fail(PESTO_ERR_INVALID, "batch too large");
    }
    // dense one-hot features -> byte indices, found while the rows are read for packing anyway: every block of every row must hold
    // exactly one 1.0f and zeros (encode_features, src/data_encoding.py:78-84); one row that does not makes the launch travel dense
    std::vector<const uint8_t*> qi_ptr;
    if (detect) {
        m->qidx_host.resize((size_t)NT * n_index);
        bool onehot = true;
        int bounds[4] = {0, 0, 0, 0};
        for (int c = 0; c < n_index; ++c) bounds[c] = index_offsets[c];
        bounds[n_index] = n0;
        for (int s_ = 0; s_ < n_struct && onehot; ++s_)       // (vectorised scans, pesto_schema.cpp)
            onehot = onehot_rows_to_indices(q0[s_], N[s_], n0, bounds, n_index, m->qidx_host.data() + (size_t)meta[s_].off * n_index);
        if (onehot) {
            qi_ptr.resize((size_t)n_struct);
            for (int s_ = 0; s_ < n_struct; ++s_) qi_ptr[s_] = m->qidx_host.data() + (size_t)meta[s_].off * n_index;
            q_index = qi_ptr.data();
        } else {
            n_index = 0;
        }
    }
    HIP_TRY(hipSetDevice(m->device));
    if (!m->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&m->copy_stream, hipStreamNonBlocking));
    if (!b.ev_h2d) { HIP_TRY(hipEventCreateWithFlags(&b.ev_h2d, hipEventDisableTiming)); HIP_TRY(hipEventCreateWithFlags(&b.ev_done, hipEventDisableTiming)); }
    if (!b.h_flag) HIP_TRY(hipHostMalloc((void**)&b.h_flag, 64, hipHostMallocDefault));
    if ((size_t)n_struct > b.hs_cap) {
        if (b.h_sflags) HIP_TRY(hipHostFree(b.h_sflags));
        b.h_sflags = nullptr; b.hs_cap = 0;
        HIP_TRY(hipHostMalloc((void**)&b.h_sflags, ((size_t)n_struct + 64) * 4, hipHostMallocDefault));
        b.hs_cap = (size_t)n_struct + 64;
    }
    // blob layout: [meta | X | ids | q (dense floats or index bytes) | roa], every part 16-byte aligned
    const size_t sz_meta = align16(meta.size() * sizeof(CollMeta));
    b.off_X = sz_meta;
    b.off_ids = b.off_X + align16((size_t)NT * 12);
    b.off_q = b.off_ids + align16((size_t)IT * id_sz);
    b.off_roa = b.off_q + align16(q_index ? (size_t)NT * n_index : (size_t)NT * n0 * 4);
    const size_t total = b.off_roa + align16((size_t)NT * 4);
    if (total > b.h_cap) {
        if (b.h_in) HIP_TRY(hipHostFree(b.h_in));
        b.h_in = nullptr; b.h_cap = 0;
        HIP_TRY(hipHostMalloc(&b.h_in, total + total / 4, hipHostMallocDefault));
        b.h_cap = total + total / 4;
    }
    const size_t zbytes = (size_t)RT * n_out * 4;
    if (zbytes > b.hz_cap) {
        if (b.h_z) HIP_TRY(hipHostFree(b.h_z));
        b.h_z = nullptr; b.hz_cap = 0;
        HIP_TRY(hipHostMalloc((void**)&b.h_z, zbytes + zbytes / 4 + 256, hipHostMallocDefault));
        b.hz_cap = zbytes + zbytes / 4 + 256;
    }
    if (int rc = ensure_workspace(m, NT, RT)) return rc;
    if (b.d_in.ensure(total) || b.d_z.ensure(zbytes) || m->col_ids.ensure((size_t)NT * KMAX * 4) || m->col_roa.ensure((size_t)NT * 4) ||
        m->col_seg.ensure((size_t)NT * 4) || m->col_segend.ensure((size_t)n_struct * 4) || (q_index && m->in_q0.ensure((size_t)NT * n0 * 4)))
        return fail(PESTO_ERR_NOMEM, "staging allocation failed");
    // pack (host): after this the caller's input arrays are free again
    char* h = (char*)b.h_in;
    memcpy(h, meta.data(), meta.size() * sizeof(CollMeta));
    for (int s_ = 0; s_ < n_struct; ++s_) {
        const CollMeta& mb = meta[s_];
        memcpy(h + b.off_X + (size_t)mb.off * 12, X[s_], (size_t)mb.n * 12);
        if (narrow) {
            if (!narrow_ids_to_u16(ids_topk0[s_], src_kind, (size_t)mb.n * mb.k, reinterpret_cast<uint16_t*>(h + b.off_ids + (size_t)mb.idoff * 2)))
                return fail(PESTO_ERR_INVALID, "structure %d: ids_topk has entries outside [0, N)", s_);
        } else {
            memcpy(h + b.off_ids + (size_t)mb.idoff * id_sz, ids_topk0[s_], (size_t)mb.n * mb.k * id_sz);
        }
        if (q_index) memcpy(h + b.off_q + (size_t)mb.off * n_index, q_index[s_], (size_t)mb.n * n_index);
        else memcpy(h + b.off_q + (size_t)mb.off * n0 * 4, q0[s_], (size_t)mb.n * n0 * 4);
        memcpy(h + b.off_roa + (size_t)mb.off * 4, res_of_atom[s_], (size_t)mb.n * 4);
    }
    b.n_struct = n_struct; b.ids_kind = ids_kind; b.n_index = q_index ? n_index : 0; b.mode = batch_mode; b.NT = NT; b.RT = RT;
    for (int c = 0; c < 3; ++c) b.index_offsets[c] = (q_index && c < n_index) ? index_offsets[c] : 0;
    if (m->host_only) {      // pesto_debug_host_only: the host half only (what one rank's CPU must sustain), nothing queued
        memset(b.h_z, 0, zbytes);
        b.h_flag[0] = b.h_flag[1] = 0;
        HIP_TRY(hipEventRecord(b.ev_done, m->stream));
        b.busy = true;
        *ticket = m->next_slot;
        m->next_slot ^= 1;
        return 0;
    }
    // the slot's device buffer may still be read by the launch that used it last: that launch was waited for (busy == false)
    HIP_TRY(hipMemcpyAsync(b.d_in.p, b.h_in, total, hipMemcpyHostToDevice, m->copy_stream));
    HIP_TRY(hipEventRecord(b.ev_h2d, m->copy_stream));
    HIP_TRY(hipStreamWaitEvent(m->stream, b.ev_h2d, 0));
    const bool exact = m->precision == PESTO_PRECISION_FP32 || m->impl != 2;
    if (int rc = queue_slot(m, b, m->stream, exact)) return rc;
    b.busy = true;
    *ticket = m->next_slot;
    m->next_slot ^= 1;
    return 0;
}

int pesto_forward_batch_wait(pesto_model* m, int32_t ticket) {
    if (check_model(m)) return PESTO_ERR_INVALID;
    if (ticket < 0 || ticket > 1 || !m->slot[ticket].busy) return fail(PESTO_ERR_STATE, "no launch in flight under ticket %d", ticket);
    pesto_model::BatchSlot& b = m->slot[ticket];
    if (int rc = resolve_pending(m)) return rc;      // (its repeat reads workspace words a repeat queued below would overwrite)
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipEventSynchronize(b.ev_done));
    struct Release { bool& busy; ~Release() { busy = false; } } release{b.busy};      // the slot is free once this call has no work queued on it
    int flag = b.h_flag[0] | b.h_flag[1];
    if (flag & 1) return fail(PESTO_ERR_INVALID, "ids_topk (or a feature index) contains an index outside its range");
    if (flag & 2) return fail(PESTO_ERR_INVALID, "res_of_atom contains an index outside [0, R)");
    if (flag & 4) {
        const bool exact_first = m->precision == PESTO_PRECISION_FP32 || m->impl != 2;
        if (m->precision != PESTO_PRECISION_AUTO || exact_first)
            return fail(PESTO_ERR_RANGE, "an activation left the f16 range of the split-MFMA path (z is NaN): use PESTO_PRECISION_AUTO or PESTO_PRECISION_FP32");
        // the slot's inputs are still on the device: the flagged structures again, on the exact fp32 kernels (the others keep their logits)
        const size_t n_words = b.mode == PESTO_BATCH_INDEPENDENT ? (size_t)b.n_struct : 1;
        for (size_t i = 0; i < n_words; ++i) m->n_rerun += (b.h_sflags[i] & 4) ? 1 : 0;
        if (int rc = queue_slot(m, b, m->stream, true, true)) { (void)hipStreamSynchronize(m->stream); return rc; }
        HIP_TRY(hipEventSynchronize(b.ev_done));
        flag = b.h_flag[0] | b.h_flag[1];
        if (flag & 3) return fail(PESTO_ERR_INVALID, "bad inputs");
    }
    const int n_out = m->cfg.n_out;
    for (int s_ = 0; s_ < b.n_struct; ++s_) memcpy(b.z_user[s_], b.h_z + (size_t)b.roff[s_] * n_out, (size_t)b.rcount[s_] * n_out * 4);
    return 0;
}

int pesto_knn_collate(pesto_model* m, int64_t n_total, int32_t n_struct, const int32_t* struct_offsets, const float* X, int32_t k,
                      void* ids_out, int32_t ids_kind, int32_t ptr_kind, void* stream) {
    if (check_model(m)) return PESTO_ERR_INVALID;
    if (int rc = resolve_pending(m)) return rc;      // a deferred AUTO repeat reads workspace words (guard words, SatCtx, segment maps) this call overwrites
    if (n_total < 1 || n_total > 0x7ffffff0 / 96 || n_struct < 1 || !struct_offsets || !X || !ids_out || k < 1 || k > KMAX)
        return fail(PESTO_ERR_INVALID, "bad arguments");
    if (ids_kind != PESTO_IDS_INT32 && ids_kind != PESTO_IDS_INT64) return fail(PESTO_ERR_INVALID, "ids_kind must be 32 or 64");
    if (struct_offsets[0] != 0 || struct_offsets[n_struct] != n_total) return fail(PESTO_ERR_INVALID, "struct_offsets must span [0, n_total]");
    for (int s = 0; s < n_struct; ++s)
        if (struct_offsets[s + 1] <= struct_offsets[s]) return fail(PESTO_ERR_INVALID, "empty or unordered structure %d", s);
    HIP_TRY(hipSetDevice(m->device));
    const size_t id_sz = ids_kind == PESTO_IDS_INT64 ? 8 : 4;
    // structures of at least knn_cell_min() atoms go through a cell grid (O(N) instead of O(N^2)); each owns one block ("slot") of
    // the cell arrays. The slot table travels behind the offsets in the same device buffer.
    std::vector<int> host_tab((size_t)2 * n_struct + 1);
    int n_slots = 0;
    const bool brute = m->knn_brute;
    for (int s = 0; s <= n_struct; ++s) host_tab[s] = struct_offsets[s];
    for (int s = 0; s < n_struct; ++s)
        host_tab[n_struct + 1 + s] = (!brute && struct_offsets[s + 1] - struct_offsets[s] >= knn_cell_min()) ? n_slots++ : -1;
    const int use_grid = n_slots > 0;
    if (m->knn_off.ensure(host_tab.size() * 4)) return fail(PESTO_ERR_NOMEM, "workspace allocation failed");
    m->knn_off_host.clear();      // (the device copy is about to change: valid again only once the upload below has been queued - ADVICE r5)
    if (use_grid) {
        const size_t cells = (size_t)n_slots * knn_cells_per_struct();
        if (m->knn_grids.ensure((size_t)n_struct * knn_grid_struct_bytes()) || m->knn_cnt.ensure(cells * 4) || m->knn_cur.ensure(cells * 4) ||
            m->knn_cell.ensure((size_t)n_total * 4) || m->knn_sorted.ensure((size_t)n_total * 16))
            return fail(PESTO_ERR_NOMEM, "workspace allocation failed");
    }
    auto knn_launch = [&](hipStream_t s_, const float* Xd, void* out) {
        launch_knn_collate(s_, (int)n_total, n_struct, m->knn_off.as<int>(), Xd, k, out, ids_kind, use_grid, m->knn_off.as<int>() + n_struct + 1,
                           m->knn_grids.p, m->knn_cnt.as<int>(),
                           m->knn_cur.as<int>(), m->knn_cell.as<int>(), m->knn_sorted.p);
    };
    if (ptr_kind == PESTO_PTR_DEVICE) {
        hipStream_t st = (hipStream_t)stream;
        Sequence seq(m, st);
        if (seq.rc) return seq.rc;
        HIP_TRY(hipMemcpyAsync(m->knn_off.p, host_tab.data(), host_tab.size() * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));     // host_tab is a local: finish the copy before returning
        m->knn_off_host.assign(struct_offsets, struct_offsets + n_struct + 1);      // (what the device holds now)
        knn_launch(st, X, ids_out);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if (ptr_kind != PESTO_PTR_HOST) return fail(PESTO_ERR_INVALID, "ptr_kind must be PESTO_PTR_HOST or PESTO_PTR_DEVICE");
    hipStream_t st = stream ? (hipStream_t)stream : m->stream;
    if (m->in_X.ensure((size_t)n_total * 12) || m->in_ids.ensure((size_t)n_total * KMAX * id_sz)) return fail(PESTO_ERR_NOMEM, "staging allocation failed");
    Sequence seq(m, st);
    if (seq.rc) return seq.rc;
    HIP_TRY(hipMemcpyAsync(m->knn_off.p, host_tab.data(), host_tab.size() * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(m->in_X.p, X, (size_t)n_total * 12, hipMemcpyHostToDevice, st));
    knn_launch(st, m->in_X.as<float>(), m->in_ids.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(ids_out, m->in_ids.p, (size_t)n_total * KMAX * id_sz, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    m->knn_off_host.assign(struct_offsets, struct_offsets + n_struct + 1);      // (the upload has completed)
    return 0;
}

int pesto_knn_tie_rows(pesto_model* m, int64_t n_total, int32_t n_struct, const int32_t* struct_offsets, const float* X, int32_t k,
                       const void* ids, int32_t ids_kind, uint8_t* flags_out, int32_t ptr_kind, void* stream) {
    if (check_model(m)) return PESTO_ERR_INVALID;
    if (int rc = resolve_pending(m)) return rc;      // a deferred AUTO repeat reads workspace words (guard words, SatCtx, segment maps) this call overwrites
    if (n_total < 1 || n_total > 0x7ffffff0 / 96 || n_struct < 1 || !struct_offsets || !X || !ids || !flags_out || k < 1 || k > KMAX)
        return fail(PESTO_ERR_INVALID, "bad arguments");
    if (ids_kind != PESTO_IDS_INT32 && ids_kind != PESTO_IDS_INT64) return fail(PESTO_ERR_INVALID, "ids_kind must be 32 or 64");
    if (ptr_kind != PESTO_PTR_HOST && ptr_kind != PESTO_PTR_DEVICE) return fail(PESTO_ERR_INVALID, "ptr_kind must be PESTO_PTR_HOST or PESTO_PTR_DEVICE");
    if (struct_offsets[0] != 0 || struct_offsets[n_struct] != n_total) return fail(PESTO_ERR_INVALID, "struct_offsets must span [0, n_total]");
    for (int s = 0; s < n_struct; ++s)
        if (struct_offsets[s + 1] <= struct_offsets[s]) return fail(PESTO_ERR_INVALID, "empty or unordered structure %d", s);
    HIP_TRY(hipSetDevice(m->device));
    const size_t id_sz = ids_kind == PESTO_IDS_INT64 ? 8 : 4;
    if (m->knn_off.cap < ((size_t)2 * n_struct + 1) * 4) m->knn_off_host.clear();      // (a reallocation drops the device copy)
    if (m->knn_off.ensure(((size_t)2 * n_struct + 1) * 4)) return fail(PESTO_ERR_NOMEM, "workspace allocation failed");
    hipStream_t st = ptr_kind == PESTO_PTR_DEVICE ? (hipStream_t)stream : (stream ? (hipStream_t)stream : m->stream);
    Sequence seq(m, st);
    if (seq.rc) return seq.rc;
    // the usual caller asks for the tie rows of the table pesto_knn_collate has just built: the same offsets are on the device already
    // (no copy, and no stream synchronisation inside a loop that is meant to stay asynchronous - ADVICE r4)
    const bool same_offsets = m->knn_off_host.size() == (size_t)n_struct + 1 &&
                              std::equal(m->knn_off_host.begin(), m->knn_off_host.end(), struct_offsets);
    if (!same_offsets) {
        HIP_TRY(hipMemcpyAsync(m->knn_off.p, struct_offsets, ((size_t)n_struct + 1) * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));           // struct_offsets is the caller's: finish the copy before returning
        m->knn_off_host.assign(struct_offsets, struct_offsets + n_struct + 1);
    }
    if (ptr_kind == PESTO_PTR_DEVICE) {
        launch_knn_ties(st, (int)n_total, n_struct, m->knn_off.as<int>(), X, k, ids, ids_kind, flags_out);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    DevBuf fl;
    if (m->in_X.ensure((size_t)n_total * 12) || m->in_ids.ensure((size_t)n_total * KMAX * id_sz) || fl.ensure((size_t)n_total))
        return fail(PESTO_ERR_NOMEM, "staging allocation failed");
    hipError_t he = hipMemcpyAsync(m->in_X.p, X, (size_t)n_total * 12, hipMemcpyHostToDevice, st);
    if (he == hipSuccess) he = hipMemcpyAsync(m->in_ids.p, ids, (size_t)n_total * KMAX * id_sz, hipMemcpyHostToDevice, st);
    if (he == hipSuccess) {
        launch_knn_ties(st, (int)n_total, n_struct, m->knn_off.as<int>(), m->in_X.as<float>(), k, m->in_ids.p, ids_kind, fl.as<unsigned char>());
        he = hipGetLastError();
    }
    if (he == hipSuccess) he = hipMemcpyAsync(flags_out, fl.p, (size_t)n_total, hipMemcpyDeviceToHost, st);
    if (he == hipSuccess) he = hipStreamSynchronize(st);
    fl.release();
    if (he != hipSuccess) return fail(PESTO_ERR_HIP, "knn_tie_rows: %s", hipGetErrorString(he));
    return 0;
}

int pesto_mask_to_segments(pesto_model* m, int64_t N, int64_t R, const float* M, int32_t* res_of_atom_out, int32_t ptr_kind, void* stream) {
    if (check_model(m)) return PESTO_ERR_INVALID;
    if (int rc = resolve_pending(m)) return rc;      // a deferred AUTO repeat reads workspace words (guard words, SatCtx, segment maps) this call overwrites
    if (N < 1 || R < 1 || N > 0x7ffffff0 / 96 || R > N || !M || !res_of_atom_out) return fail(PESTO_ERR_INVALID, "bad arguments");
    if (ptr_kind != PESTO_PTR_HOST && ptr_kind != PESTO_PTR_DEVICE) return fail(PESTO_ERR_INVALID, "ptr_kind must be PESTO_PTR_HOST or PESTO_PTR_DEVICE");
    HIP_TRY(hipSetDevice(m->device));
    const size_t cap0 = m->mask_seen.cap;
    if (m->mask_seen.ensure((size_t)R * 4)) return fail(PESTO_ERR_NOMEM, "workspace allocation failed");
    if (m->mask_seen.cap != cap0 || m->mask_gen == 0x7fffffff) m->mask_seen_clean = false;      // a fresh buffer holds anything: cleared once (on the call's stream), generations restart
    const bool fresh = !m->mask_seen_clean;
    if (fresh) m->mask_gen = 0;
    const int gen = ++m->mask_gen;
    if (ptr_kind == PESTO_PTR_DEVICE) {
        Sequence seq(m, (hipStream_t)stream);
        if (seq.rc) return seq.rc;
        if (fresh) { HIP_TRY(hipMemsetAsync(m->mask_seen.p, 0, m->mask_seen.cap, (hipStream_t)stream)); m->mask_seen_clean = true; }
        launch_mask_to_segments((hipStream_t)stream, (int)N, (int)R, M, res_of_atom_out, m->mask_seen.as<int>(), gen);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    hipStream_t st = stream ? (hipStream_t)stream : m->stream;
    if (m->in_M.ensure((size_t)N * R * 4) || m->in_roa.ensure((size_t)N * 4)) return fail(PESTO_ERR_NOMEM, "staging allocation failed");
    Sequence seq(m, st);
    if (seq.rc) return seq.rc;
    HIP_TRY(hipMemcpyAsync(m->in_M.p, M, (size_t)N * R * 4, hipMemcpyHostToDevice, st));
    if (fresh) { HIP_TRY(hipMemsetAsync(m->mask_seen.p, 0, m->mask_seen.cap, st)); m->mask_seen_clean = true; }
    launch_mask_to_segments(st, (int)N, (int)R, m->in_M.as<float>(), m->in_roa.as<int>(), m->mask_seen.as<int>(), gen);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(res_of_atom_out, m->in_roa.p, (size_t)N * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (int64_t i = 0; i < N; ++i)
        if (res_of_atom_out[i] < 0)
            return fail(PESTO_ERR_INVALID, i == 0 ? "M: atom 0 has != 1 residue, or a residue column is empty" : "M: atom %lld belongs to != 1 residue", (long long)i);
    return 0;
}

int pesto_postprocess(pesto_model* m, int64_t N, int64_t R, const float* z, const int32_t* res_of_atom, float* p_out, float* bfactor_out,
                      int32_t ptr_kind, void* stream) {
    if (check_model(m)) return PESTO_ERR_INVALID;
    if (int rc = resolve_pending(m)) return rc;       // (the usual first consumer of an asynchronous forward's logits)
    if (N < 1 || R < 1 || N > 0x7ffffff0 / 96 || !z || (!p_out && !bfactor_out) || (bfactor_out && !res_of_atom)) return fail(PESTO_ERR_INVALID, "bad arguments");
    HIP_TRY(hipSetDevice(m->device));
    const int n_out = m->cfg.n_out;
    if (m->flags.ensure(64)) return fail(PESTO_ERR_NOMEM, "workspace allocation failed");
    if (ptr_kind == PESTO_PTR_DEVICE) {
        // no flag reset / read-back here: the call stays asynchronous; a bad res_of_atom entry is clamped to residue 0
        Sequence seq(m, (hipStream_t)stream);
        if (seq.rc) return seq.rc;
        launch_postprocess((hipStream_t)stream, (int)N, (int)R, n_out, z, res_of_atom, p_out, bfactor_out, err_ptr(m));
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if (ptr_kind != PESTO_PTR_HOST) return fail(PESTO_ERR_INVALID, "ptr_kind must be PESTO_PTR_HOST or PESTO_PTR_DEVICE");
    hipStream_t st = stream ? (hipStream_t)stream : m->stream;
    DevBuf out;
    if (m->z.ensure((size_t)R * 32 * 4) || m->in_roa.ensure((size_t)N * 4) || out.ensure(((size_t)R + N) * n_out * 4))
        return fail(PESTO_ERR_NOMEM, "staging allocation failed");
    float* dp = out.as<float>();
    float* db = dp + (size_t)R * n_out;
    Sequence seq(m, st);
    if (seq.rc) { out.release(); return seq.rc; }
    int rc = 0;
    do {
        hipError_t he = hipMemsetAsync(m->flags.p, 0, 8, st);
        auto also = [&he](hipError_t e) { if (he == hipSuccess) he = e; };
        also(hipMemcpyAsync(m->z.p, z, (size_t)R * n_out * 4, hipMemcpyHostToDevice, st));
        if (bfactor_out) also(hipMemcpyAsync(m->in_roa.p, res_of_atom, (size_t)N * 4, hipMemcpyHostToDevice, st));
        launch_postprocess(st, (int)N, (int)R, n_out, m->z.as<float>(), m->in_roa.as<int>(), dp, bfactor_out ? db : nullptr, err_ptr(m));
        also(hipGetLastError());
        if (p_out) also(hipMemcpyAsync(p_out, dp, (size_t)R * n_out * 4, hipMemcpyDeviceToHost, st));
        if (bfactor_out) also(hipMemcpyAsync(bfactor_out, db, (size_t)N * n_out * 4, hipMemcpyDeviceToHost, st));
        if (he != hipSuccess) { rc = fail(PESTO_ERR_HIP, "postprocess: %s", hipGetErrorString(he)); break; }
        rc = check_device_flag(m, st);
    } while (0);
    out.release();
    return rc;
}

// ------------------------------------------------------------------ per-stage entry points (host pointers)
int pesto_stage_embed(pesto_model* m, int64_t N, const float* q0, float* q_out) {
    if (check_model(m)) return PESTO_ERR_INVALID;
    if (int rc = resolve_pending(m)) return rc;      // a deferred AUTO repeat reads workspace words (guard words, SatCtx, segment maps) this call overwrites
    if (N < 1 || !q0 || !q_out) return fail(PESTO_ERR_INVALID, "bad arguments");
    HIP_TRY(hipSetDevice(m->device));
    if (int rc = ensure_workspace(m, N, 1)) return rc;
    if (m->in_q0.ensure((size_t)N * m->cfg.n0 * 4)) return fail(PESTO_ERR_NOMEM, "staging allocation failed");
    hipStream_t st = m->stream;
    Sequence seq(m, st);
    if (seq.rc) return seq.rc;
    HIP_TRY(hipMemcpyAsync(m->in_q0.p, q0, (size_t)N * m->cfg.n0 * 4, hipMemcpyHostToDevice, st));
    launch_embed(st, m->W, m->img.model.em, (int)N, (int)N, m->cfg.n0, m->in_q0.as<float>(), m->q_a.as<float>());
    HIP_TRY(hipMemcpyAsync(q_out, m->q_a.as<float>() + S, (size_t)N * S * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    m->stage_N = -1;
    return 0;
}

int pesto_stage_unpack(pesto_model* m, int64_t N, int32_t k, const float* X, const void* ids_topk, int32_t ids_kind,
                       float* D_out, float* R_out) {
    if (check_model(m)) return PESTO_ERR_INVALID;
    if (int rc = resolve_pending(m)) return rc;      // a deferred AUTO repeat reads workspace words (guard words, SatCtx, segment maps) this call overwrites
    if (N < 1 || k < 1 || k > KMAX || !X || !ids_topk) return fail(PESTO_ERR_INVALID, "bad arguments");
    if (ids_kind != PESTO_IDS_INT32 && ids_kind != PESTO_IDS_INT64) return fail(PESTO_ERR_INVALID, "ids_kind must be 32 or 64");
    HIP_TRY(hipSetDevice(m->device));
    if (int rc = ensure_workspace(m, N, 1)) return rc;
    const size_t id_sz = ids_kind == PESTO_IDS_INT64 ? 8 : 4;
    if (m->in_X.ensure((size_t)N * 3 * 4) || m->in_ids.ensure((size_t)N * k * id_sz)) return fail(PESTO_ERR_NOMEM, "staging allocation failed");
    hipStream_t st = m->stream;
    if (m->dmax.ensure(4)) return fail(PESTO_ERR_NOMEM, "workspace allocation failed");
    Sequence seq(m, st);
    if (seq.rc) return seq.rc;
    HIP_TRY(hipMemsetAsync(m->flags.p, 0, 8, st));
    HIP_TRY(hipMemsetAsync(m->dmax.p, 0, 4, st));
    HIP_TRY(hipMemcpyAsync(m->in_X.p, X, (size_t)N * 3 * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(m->in_ids.p, ids_topk, (size_t)N * k * id_sz, hipMemcpyHostToDevice, st));
    launch_unpack(st, (int)N, 1, k, m->in_X.as<float>(), 3 * N, 3, m->in_ids.p, ids_kind, m->ids_s.as<int>(), m->geo.as<float4>(), dmax_ptr(m),
                  err_ptr(m));
    if (int rc = check_device_flag(m, st)) return rc;
    m->stage_N = N;
    if (D_out || R_out) {
        std::vector<float4> g((size_t)(N + 1) * KMAX);
        HIP_TRY(hipMemcpy(g.data(), m->geo.p, g.size() * sizeof(float4), hipMemcpyDeviceToHost));
        for (int64_t i = 0; i <= N; ++i)
            for (int c = 0; c < k; ++c) {
                const float4 v = g[(size_t)i * KMAX + c];
                if (D_out) D_out[i * k + c] = v.w;
                if (R_out) { R_out[(i * k + c) * 3] = v.x; R_out[(i * k + c) * 3 + 1] = v.y; R_out[(i * k + c) * 3 + 2] = v.z; }
            }
    }
    return 0;
}

int pesto_stage_layer(pesto_model* m, int32_t layer, float* q_io, float* p_io) {
    if (check_model(m)) return PESTO_ERR_INVALID;
    if (int rc = resolve_pending(m)) return rc;      // a deferred AUTO repeat reads workspace words (guard words, SatCtx, segment maps) this call overwrites
    if (m->stage_N < 1) return fail(PESTO_ERR_STATE, "pesto_stage_unpack must run first");
    if (layer < 0 || layer >= m->cfg.n_layers || !q_io || !p_io) return fail(PESTO_ERR_INVALID, "bad arguments");
    HIP_TRY(hipSetDevice(m->device));
    const size_t N1 = (size_t)m->stage_N + 1;
    hipStream_t st = m->stream;
    Sequence seq(m, st);
    if (seq.rc) return seq.rc;
    if (m->sflags.ensure(4)) return fail(PESTO_ERR_NOMEM, "workspace allocation failed");
    HIP_TRY(hipMemsetAsync(m->flags.p, 0, 8, st));
    HIP_TRY(hipMemsetAsync(m->sflags.p, 0, 4, st));
    const SatCtx sc{err_ptr(m), m->sflags.as<int>(), nullptr, 0};      // (no embed launch here: the guard's context is uploaded)
    HIP_TRY(hipMemcpyAsync(err_ptr(m) + SATCTX_OFFSET_INTS, &sc, sizeof sc, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipMemcpyAsync(m->q_a.p, q_io, N1 * S * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(m->p_a.p, p_io, N1 * 96 * 4, hipMemcpyHostToDevice, st));
    const void *q_res = m->q_b.p, *p_res = m->p_b.p;
    if (m->impl == 2) {
        const LayerW* L = &m->img.layers[layer];
        const int ev = m->precision == PESTO_PRECISION_FP32 ? 1 : 0;     // no automatic re-run at stage level
        launch_node(st, m->W, nullptr, L, (int)N1, m->q_a.as<float>(), m->p_a.as<float>(), m->zrec.as<float>(), m->rec_nb.as<float>(), m->rec_cen.as<float>(), ev, err_ptr(m));
        if (ev == 0) {      // shipped path: the finish phase runs inside the edge kernel, new state in the other buffer pair
            launch_edge(st, m->W, *L, (int)N1, m->ids_s.as<int>(), m->geo.as<float4>(), m->rec_nb.as<float>(), m->rec_cen.as<float>(), m->p_a.as<float>(), m->zrec.as<float>(), m->edge_blocks, ev, err_ptr(m),
                        m->q_a.as<float>(), m->q_b.as<float>(), m->p_b.as<float>(), nullptr, nullptr, nullptr, m->edge_mode);
        } else {
            launch_edge(st, m->W, *L, (int)N1, m->ids_s.as<int>(), m->geo.as<float4>(), m->rec_nb.as<float>(), m->rec_cen.as<float>(), m->p_a.as<float>(), m->zrec.as<float>(), m->edge_blocks, ev, err_ptr(m));
            launch_node(st, m->W, L, nullptr, (int)N1, m->q_a.as<float>(), m->p_a.as<float>(), m->zrec.as<float>(), m->rec_nb.as<float>(), m->rec_cen.as<float>(), ev, err_ptr(m));
            q_res = m->q_a.p; p_res = m->p_a.p;
        }
    } else {
        launch_layer_v1(st, m->W, m->img.layers[layer], (int)N1, m->ids_s.as<int>(), m->geo.as<float4>(), m->q_a.as<float>(), m->p_a.as<float>(),
                        m->q_b.as<float>(), m->p_b.as<float>());
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(q_io, q_res, N1 * S * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(p_io, p_res, N1 * 96 * 4, hipMemcpyDeviceToHost, st));
    return check_device_flag(m, st);      // synchronises; reports a range overflow of the f16-split kernels (PESTO_ERR_RANGE)
}

int pesto_stage_pool(pesto_model* m, int64_t N, int64_t R, const float* q, const float* p, const int32_t* res_of_atom,
                     float* qr_out, float* pr_out, float* z_out) {
    if (check_model(m)) return PESTO_ERR_INVALID;
    if (int rc = resolve_pending(m)) return rc;      // a deferred AUTO repeat reads workspace words (guard words, SatCtx, segment maps) this call overwrites
    if (N < 1 || R < 1 || !q || !p || !res_of_atom || !z_out) return fail(PESTO_ERR_INVALID, "bad arguments");
    HIP_TRY(hipSetDevice(m->device));
    if (int rc = ensure_workspace(m, N, R)) return rc;
    if (m->in_roa.ensure((size_t)N * 4)) return fail(PESTO_ERR_NOMEM, "staging allocation failed");
    DevBuf qr, pr;
    if (qr.ensure((size_t)R * S * 4) || pr.ensure((size_t)R * 96 * 4)) return fail(PESTO_ERR_NOMEM, "staging allocation failed");
    hipStream_t st = m->stream;
    Sequence seq(m, st);
    if (seq.rc) { qr.release(); pr.release(); return seq.rc; }
    int rc = 0;
    do {
        hipError_t he = hipMemsetAsync(m->flags.p, 0, 8, st);
        auto also = [&he](hipError_t e) { if (he == hipSuccess) he = e; };
        also(hipMemcpyAsync(m->q_a.as<float>() + S, q, (size_t)N * S * 4, hipMemcpyHostToDevice, st));
        also(hipMemcpyAsync(m->p_a.as<float>() + 96, p, (size_t)N * 96 * 4, hipMemcpyHostToDevice, st));
        also(hipMemcpyAsync(m->in_roa.p, res_of_atom, (size_t)N * 4, hipMemcpyHostToDevice, st));
        launch_pool(st, m->W, m->img.model, m->cfg.n_out, (int)N, (int)R, m->q_a.as<float>() + S, m->p_a.as<float>() + 96, m->in_roa.as<int>(),
                    m->pool_a.as<float>(), m->seg.as<int>(), m->seg.as<int>() + R, err_ptr(m), qr.as<float>(), pr.as<float>(), m->z.as<float>());
        also(hipGetLastError());
        if (qr_out) also(hipMemcpyAsync(qr_out, qr.p, (size_t)R * S * 4, hipMemcpyDeviceToHost, st));
        if (pr_out) also(hipMemcpyAsync(pr_out, pr.p, (size_t)R * 96 * 4, hipMemcpyDeviceToHost, st));
        also(hipMemcpyAsync(z_out, m->z.p, (size_t)R * m->cfg.n_out * 4, hipMemcpyDeviceToHost, st));
        if (he != hipSuccess) { rc = fail(PESTO_ERR_HIP, "stage_pool: %s", hipGetErrorString(he)); break; }
        rc = check_device_flag(m, st);
    } while (0);
    qr.release(); pr.release();
    m->stage_N = -1;
    return rc;
}

}  // extern "C"
