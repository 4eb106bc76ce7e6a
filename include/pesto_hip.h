/* pesto_hip.h - C ABI of libpesto_hip.so: PeSTo's geometric-transformer forward pass on MI355X (gfx950).
 *
 * The reference has NO native/FFI layer: its operator API for this path is the torch Module
 *     Model(config_model).load_state_dict(...); z = model(X, ids_topk, q0, M)
 * (reference model/model.py:7-52; call sites apply_model.ipynb:87-93,155, profiling.py:24-28,102,
 * interfaceome/apply_model.py:33-34,73).  Each entry point below names the reference interface it
 * replaces.  Plain pointers and sizes only; no torch types.  All functions return 0 on success and a
 * negative pesto_status on failure; pesto_last_error() gives the thread-local message.
 *
 * Conventions (identical to the reference tensors):
 *   X          float32 [N,3]        atom coordinates of the collated batch
 *   ids_topk   int64 or int32 [N,k] 1-based neighbour ids into the sink-augmented arrays, 0 = sink /
 *                                   padding, columns in ascending distance (k <= 64)
 *   q0         float32 [N,n0]       input features (one-hot in practice, any values accepted)
 *   res_of_atom int32 [N]           residue (column of the reference's mask M) each atom belongs to;
 *                                   the host layer derives it from M (exactly one 1 per row)
 *   z          float32 [R,n_out]    logits per residue
 */
#ifndef PESTO_HIP_H
#define PESTO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PESTO_MAX_LAYERS 64
#define PESTO_MAX_K 64

typedef enum pesto_status {
    PESTO_OK = 0,
    PESTO_ERR_INVALID = -1,   /* bad argument / config / blob size */
    PESTO_ERR_HIP = -2,       /* a HIP runtime call failed */
    PESTO_ERR_NOMEM = -3,
    PESTO_ERR_STATE = -4,     /* debug entry point called out of order */
    PESTO_ERR_RANGE = -5      /* PESTO_PRECISION_F16_SPLIT only: an activation left the f16 range (z has been filled with NaN) */
} pesto_status;

/* Arithmetic of the state-update layers (the reference computes in fp32 throughout, src/model_operations.py:87-154).
 *   F16_SPLIT : every large GEMM as three v_mfma_f32_16x16x32_f16 products of f16 hi/lo pairs (x = hi + lo), fp32 accumulate:
 *               ~22-bit mantissa, but the f16 EXPONENT range - activations beyond +-65504 cannot be represented. The kernels
 *               detect that (range guard); the result is then NaN everywhere and, where the call synchronises, PESTO_ERR_RANGE.
 *   FP32      : everything on exact fp32 MFMA (v_mfma_f32_16x16x4_f32), no range limit, about half the speed.
 *   AUTO      : F16_SPLIT, and every STRUCTURE whose range guard fired is computed again on the FP32 kernels (the reference's trained
 *               i_v3_1, model/save/i_v3_1_2021-05-28_12-40, needs this: its states reach 4e5). Default.
 *               The guard is kept per structure of a launch (one word per member of a PESTO_BATCH_INDEPENDENT batch / per trajectory
 *               frame / per collated call): the repeat runs the exact kernels over the launch again but writes only the logits of
 *               the flagged structures; the others keep the logits of the split kernels. A structure's bits therefore do not depend
 *               on what shared its launch - bitwise independence of the grouping holds under AUTO as under F16_SPLIT / FP32 -
 *               and not on the handle's history either (no "fp32 first after one overflow" switch: a model that overflows on every
 *               input pays the repeat every time; pesto_get_status counts the structures repeated - use PESTO_PRECISION_FP32 for it).
 *               Host-pointer calls (which wait for their D2H copy anyway) repeat before they return. Device-pointer calls also
 *               CHECK BEFORE THEY RETURN: the flags word is final behind the last layer launch, is copied (4 bytes) in front of the
 *               pool kernels, and the call waits for THAT copy - not for the stream. When the call returns the verdict is known (bad
 *               inputs have raised, an overflowed structure has been queued again on the exact kernels), but the pool kernels may
 *               still be reading res_of_atom and writing z_out: z_out is final IN STREAM ORDER on `stream` (what a drop-in caller
 *               that goes on with further work on that stream needs), and the input / output buffers must stay valid - and must not
 *               be overwritten from the host or from another stream - until `stream` has passed the call (pesto_synchronize, or a
 *               stream / event wait of the caller's). pesto_set_async_auto(m, 1) trades the check for a fully asynchronous
 *               call: the flags word is copied to pinned memory behind the launch and looked at by the NEXT call on the handle (any
 *               forward, pesto_postprocess, pesto_get_status, pesto_set_precision, pesto_forward_batch_wait, pesto_synchronize):
 *               bad inputs are reported there (PESTO_ERR_INVALID), a range overflow queues the fp32 repeat of the remembered launch
 *               into the same z_out - the caller keeps the buffers of such a call valid until that next call / pesto_synchronize
 *               returns; until then the structures of an overflowed launch hold NaN logits (loud, never a plausible wrong number). */
typedef enum pesto_precision {
    PESTO_PRECISION_AUTO = 0,
    PESTO_PRECISION_F16_SPLIT = 1,
    PESTO_PRECISION_FP32 = 2
} pesto_precision;

/* replaces: the config dict consumed by Model.__init__ (model/model.py:7-30, model/config.py:25-63).
 * Ns=32, Nh=2, Nk=3, pool Nh=4, N1=32 are fixed (true for every run shipped with the reference). */
typedef struct pesto_config {
    int32_t n0;                    /* config["em"]["N0"]: 30 (i_v4_*) or 123 (i_v3_*) */
    int32_t n_layers;              /* len(config["sum"]) */
    int32_t nn[PESTO_MAX_LAYERS];  /* config["sum"][l]["nn"] in {8,16,32,64} */
    int32_t n_out;                 /* config["dm"]["N2"] */
    int32_t em_depth;              /* 3 = Linear-ELU-Linear-ELU-Linear (model/model.py:10-16); 1 = single Linear (i_v3_1) */
    int32_t dm_depth;              /* same for the decoder (model/model.py:24-30) */
    int32_t precision;             /* pesto_precision; no reference counterpart (torch computes in fp32) */
} pesto_config;

typedef struct pesto_model pesto_model;

enum { PESTO_PTR_HOST = 0, PESTO_PTR_DEVICE = 1 };
enum { PESTO_IDS_INT32 = 32, PESTO_IDS_INT64 = 64, PESTO_IDS_UINT16 = 16 /* pesto_forward_batch_submit only */,
       PESTO_IDS_NARROW = 0x100 /* pesto_forward_batch_submit only, OR-ed to INT32 / INT64: stage the table as uint16 (narrowed and range-checked
                                   while it is packed) when every structure has <= 65,536 atoms */ };

const char* pesto_last_error(void);

/* number of float32 values in the weight blob for this config (schema: pesto_amd/weights.py) */
int pesto_blob_size(const pesto_config* cfg, int64_t* n_floats);

/* replaces: Model(config) + load_state_dict + .to(device)  (apply_model.ipynb:87-93).
 * weights: HOST pointer to the flat blob (state_dict order, m_nn/sdk skipped). The library keeps its
 * own device copy (re-laid-out for the kernels). device: HIP device ordinal. */
int pesto_create(const pesto_config* cfg, const float* weights, int64_t n_weights, int device, pesto_model** out);
int pesto_destroy(pesto_model* m);

/* change the precision policy of an existing handle (takes effect with the next call) / read it back together with the number
 * of launch sequences run so far and how many STRUCTURES AUTO computed again on the fp32 kernels */
int pesto_set_precision(pesto_model* m, int32_t precision);
int pesto_get_status(const pesto_model* m, int32_t* precision, int64_t* n_forward, int64_t* n_fp32_rerun);
/* PESTO_PRECISION_AUTO's bill: structures (members of a batch, trajectory frames, collated calls) forwarded on the split kernels and how
 * many of them were computed again on the exact fp32 kernels. A model whose ratio stays near 1 - the reference's trained i_v3_1
 * (model/save/i_v3_1_2021-05-28_12-40: states of 4e5 from layer 13 on) repeats EVERY structure - pays split + exact on every call and
 * belongs on PESTO_PRECISION_FP32; the Python layer logs that once (logging "pesto_amd", WARNING). No reference counterpart. */
int pesto_get_auto_counters(const pesto_model* m, int64_t* n_structures, int64_t* n_repeated);
/* enabled != 0: device-pointer forwards under PESTO_PRECISION_AUTO return without synchronising; their range / input check is made by
 * the next call on the handle (see pesto_precision). Default 0: checked before the call returns. No reference counterpart. */
int pesto_set_async_auto(pesto_model* m, int32_t enabled);
/* Conditioning trigger of PESTO_PRECISION_AUTO (round 5; no reference counterpart - torch computes in fp32 throughout,
 * src/model_operations.py:109-152). The f16 hi/lo split drops a term of 2^-22 RELATIVE size per product, so the absolute error of the
 * logits grows with the magnitude of the states. A structure whose new state exceeds `limit` (max |q|, |p| of any atom) in any layer is
 * flagged like a range overflow and repeated on the exact fp32 kernels (counted by pesto_get_status). limit <= 0 switches the trigger
 * off; F16_SPLIT and FP32 ignore it.
 * Calibration (profiles/r05_state_limit.txt): the states of REAL structures reach 48 - 96 on a few of the 53 pdbs_test chains (three
 * chains above 48, one above 64 with the i_v4_1 architecture, none above 96) while their logits stay within 3e-5 of the reference, and the
 * two pinned ill-conditioned random clouds (tests/golden/fuzz_pins.npz, |p| ~ 50) sit BELOW that - state magnitude does not separate
 * them, so the default is a safety net between the sizes trained models produce and the f16 range (65,504), not a tuned detector:
 * 128 fires on none of the reference's test structures. (The pinned inputs themselves are at 8.3e-5 / 8.2e-5 from the reference's fp64
 * logits with the trigger off; a caller that wants the exact kernels on such inputs sets a lower limit or PESTO_PRECISION_FP32.) */
#define PESTO_AUTO_STATE_LIMIT_DEFAULT 128.0f
int pesto_set_auto_state_limit(pesto_model* m, float limit);
/* Second conditioning trigger of PESTO_PRECISION_AUTO, on by default: a structure with ZERO-PADDED neighbour slots - fewer than 64 atoms,
 * or a table of fewer than 64 columns (collate_batch_features pads with 0, src/dataset.py:100-109; unpack_state_features wraps those slots
 * to the last atom, src/model_operations.py:8) - is repeated on the exact fp32 kernels. Padded slots are where the forward is
 * ill-conditioned for any fp32 evaluation (the reference's own fp32 and fp64 runs differ by 1.15e-4 on the pinned 2-atom member of
 * tests/golden/fuzz_pins.npz; the split kernels by 0.8 - 1.4e-4 depending on summation order, the exact kernels by 2e-5); real structures
 * have none, so the repeat costs nothing where throughput matters. The repeat of AUTO runs the exact kernels over the WHOLE launch of a
 * flagged structure (it writes only that structure's logits): bulk callers give such structures launches of their own
 * (pesto_amd.sharding.forward_local and pesto_amd.apply.apply_model do). enabled == 0 switches it off. F16_SPLIT / FP32 ignore it. */
int pesto_set_auto_pad_trigger(pesto_model* m, int32_t enabled);

/* replaces: Model.forward(X, ids_topk, q0, M)  (model/model.py:32-52).
 * ptr_kind: PESTO_PTR_HOST (library stages H2D/D2H itself) or PESTO_PTR_DEVICE (all five buffers on
 * the model's device). stream: a hipStream_t. With device pointers the work is queued on exactly that stream
 * (NULL = HIP's default stream, which is what torch.cuda.current_stream() is by default). Under PESTO_PRECISION_AUTO the flags word
 * (bad ids / residue columns, range overflow) is read back behind the launch and the call returns once it has been checked
 * (pesto_set_async_auto(m, 1): returns at once, checked by the next call on the handle - see pesto_precision above); under
 * F16_SPLIT / FP32 the call returns at once and nothing is checked (bad inputs make every logit NaN, an overflow its structure's).
 * With host pointers NULL selects the model's own stream, the call returns after z has been copied back, and the checks
 * (PESTO_ERR_INVALID, the fp32 repeat under AUTO, PESTO_ERR_RANGE under F16_SPLIT) happen before it returns.
 * A handle owns ONE workspace: calls on different streams are ordered after one another through an event, never concurrent.
 * No allocation happens on this path once the grow-only workspace has seen a batch of this size. */
int pesto_forward(pesto_model* m, int64_t N, int64_t R, int32_t k,
                  const float* X, const void* ids_topk, int32_t ids_kind,
                  const float* q0, const int32_t* res_of_atom,
                  float* z_out, int32_t ptr_kind, void* stream);

/* replaces: the reference's bulk inference loops, which call Model.forward ONCE PER STRUCTURE (apply_model.ipynb:139-167,
 * interfaceome/apply_model.py:57-82, profiling.py:84-108), for n_struct structures laid out as one collated batch
 * (X, ids_topk, q0, res_of_atom exactly as for pesto_forward; ids_topk e.g. from pesto_knn_collate with the same offsets).
 * Structure s owns atoms [struct_offsets[s], struct_offsets[s+1]) (HOST array, n_struct + 1 entries). One launch sequence
 * for the whole batch, but the two places where the reference's forward couples the atoms of a call - the wrap-around of
 * zero-padded neighbour slots to the LAST atom and the global max(D) of the coincident-atom fix-up
 * (src/model_operations.py:8-12) - act per structure, so every structure gets the result of its own call, independent
 * of its batch mates. Pointer / stream / precision rules as pesto_forward. */
int pesto_forward_structures(pesto_model* m, int64_t N, int64_t R, int32_t k, int32_t n_struct, const int32_t* struct_offsets,
                             const float* X, const void* ids_topk, int32_t ids_kind,
                             const float* q0, const int32_t* res_of_atom,
                             float* z_out, int32_t ptr_kind, void* stream);

/* replaces: the per-frame loop of the reference's MD analysis (md_analysis/apply_model_md.ipynb cell 6):
 *     for i in frames: z_i = model(X_traj[:, i], ids_topk, q, M)        # ids_topk, q, M of frame 0 for every frame
 * n_frames coordinate sets of the SAME N atoms share ids_topk [N,k], q0 [N,n0] and res_of_atom [N]; frame f's atom i is at
 * X + f*x_frame_stride + i*x_atom_stride (strides in floats, xyz contiguous; the reference's [N, frames, 3] trajectory
 * tensor is x_frame_stride = 3, x_atom_stride = 3*frames). z_out is [n_frames, R, n_out]. Results are those of n_frames
 * separate pesto_forward calls (per-frame max(D) and wrap-around, src/model_operations.py:8-12), but frames_per_launch
 * frames (0 = choose: about 24.6k atoms, four full rounds of the layer kernels) run as ONE batch through every kernel. Pointer/stream rules as pesto_forward. */
int pesto_forward_frames(pesto_model* m, int64_t N, int64_t R, int32_t k, int64_t n_frames,
                         const float* X, int64_t x_frame_stride, int64_t x_atom_stride,
                         const void* ids_topk, int32_t ids_kind, const float* q0, const int32_t* res_of_atom,
                         float* z_out, int32_t frames_per_launch, int32_t ptr_kind, void* stream);

/* replaces: collate_batch_features (src/dataset.py:91-112) + Model.forward for a LIST of structures - the bulk drivers' loop
 * (apply_model.ipynb:139-167, interfaceome/apply_model.py:57-82) with several structures per launch.
 * HOST pointers, one entry per structure b: X[b] float32 [N_b,3]; ids_topk0[b] [N_b,k_b] 0-BASED within the structure, as
 * extract_topology returns them (k_b = min(64, N_b)); q0[b] [N_b,n0]; res_of_atom[b] int32 [N_b] (column of the structure's
 * own mask, R_b residues); z_out[b] float32 [R_b,n_out]. The arrays are copied back to back and collated ON THE DEVICE
 * (offset + 1-based ids zero-padded to 64 columns, residue columns offset). batch_mode:
 *   PESTO_BATCH_COLLATED    exactly the reference's forward on the collated batch (training-style batches, src/dataset.py:91-112):
 *                           ONE max(D) for the whole batch and zero-padded neighbour slots wrap to the LAST atom of the batch
 *                           (src/model_operations.py:8-12) - a structure's result depends on its batch mates when it has fewer
 *                           than 64 atoms or coincident atoms;
 *   PESTO_BATCH_INDEPENDENT each structure as in its own call, which is what the reference's bulk inference loops do (one
 *                           structure per forward): max(D) and the wrap target are per structure, so results do not depend on
 *                           how structures were grouped into launches (what sharding over GPUs relies on) - bit for bit under
 *                           every precision: under AUTO the range guard and its fp32 repeat are per structure (pesto_precision).
 * Returns after every z_out[b] is filled. */
enum { PESTO_BATCH_COLLATED = 0, PESTO_BATCH_INDEPENDENT = 1 };
int pesto_forward_batch(pesto_model* m, int32_t n_struct, const int64_t* N, const int64_t* R, const int32_t* k,
                        const float* const* X, const void* const* ids_topk0, int32_t ids_kind, const float* const* q0,
                        const int32_t* const* res_of_atom, float* const* z_out, int32_t batch_mode, void* stream);

/* replaces: the same bulk loop fed by DataLoader(num_workers = 8) (interfaceome/apply_model.py:50-82; SURVEY 8b, threading row: "host may
 * overlap H2D of structure t+1 with compute of t"). pesto_forward_batch in two halves, with TWO staging slots per handle:
 *   submit  packs the structures into the slot's PINNED host buffer, queues one H2D copy on the handle's copy stream and - behind it, on
 *           the handle's compute stream - the collate kernel, the forward and the D2H copy of the logits into pinned memory; returns at
 *           once with the slot as *ticket. A second submit may follow before the first is waited for: its packing and H2D copy overlap
 *           the first launch's kernels (PESTO_ERR_STATE if both slots are in flight).
 *   wait    blocks until the ticket's logits have arrived, reports bad inputs (PESTO_ERR_INVALID) / handles the range guard as
 *           pesto_forward_batch does (AUTO: the slot's inputs are still on the device, the launch is repeated on the fp32 kernels),
 *           and copies the logits into the z_out[b] arrays given at submit (which must stay valid until then; the input arrays may be
 *           reused as soon as submit returns).
 * Inputs as pesto_forward_batch, plus two compact forms that cut the H2D volume from 392 to 145 bytes per atom:
 *   ids_kind PESTO_IDS_UINT16  ids_topk0[b] as uint16 (0-based within the structure, N_b <= 65,536);
 *   q_index != NULL            instead of the dense one-hot q0[b]: uint8 [N_b, n_index] block-local indices, expanded on the GPU to
 *                              q0[i][index_offsets[c] + q_index[i][c]] = 1 (encode_features, src/data_encoding.py:78-84);
 *   q_index == NULL, n_index > 0   dense q0[b] AND the block offsets: the library looks for the indices itself while it reads the rows
 *                              for packing (every block of every row exactly one 1.0f, everything else 0.0f) and ships them as bytes; a
 *                              launch with one row that is not one-hot travels dense - same bits either way. */
int pesto_forward_batch_submit(pesto_model* m, int32_t n_struct, const int64_t* N, const int64_t* R, const int32_t* k,
                               const float* const* X, const void* const* ids_topk0, int32_t ids_kind,
                               const float* const* q0, const uint8_t* const* q_index, int32_t n_index, const int32_t* index_offsets,
                               const int32_t* const* res_of_atom, float* const* z_out, int32_t batch_mode, int32_t* ticket);
int pesto_forward_batch_wait(pesto_model* m, int32_t ticket);
/* test / measurement hook: enabled != 0 makes pesto_forward_batch_submit do its HOST half only (validation, one-hot detection, packing
 * into the pinned slot) and queue nothing on the GPU; the wait returns zero logits. profiles/host_packing.py uses it to measure the
 * packing rate one rank's CPU sustains with the GPU idle. */
int pesto_debug_host_only(pesto_model* m, int32_t enabled);

/* bytes of device workspace a batch of (N, R) needs (ownership: SURVEY 8b) */
int pesto_workspace_bytes(const pesto_model* m, int64_t N, int64_t R, int64_t* bytes);

/* wait for everything queued on the model's own stream; also runs the deferred check of the last asynchronous AUTO launch
 * (pesto_set_async_auto; its error code is returned; the structures of an overflowed launch are repeated on the fp32 kernels and waited for) */
int pesto_synchronize(pesto_model* m);

/* mean duration in milliseconds of the state-update kernels of the most recent pesto_forward, measured with
 * HIP events on the stream they ran on (bench.py's roofline leg); enable with pesto_set_timing(m, 1). */
int pesto_set_timing(pesto_model* m, int32_t enabled);
int pesto_get_timing(pesto_model* m, double* layers_ms, double* total_ms, int32_t* n_layer_launches);
/* pesto_set_timing(m, 2) additionally records one event between consecutive layer launches; this returns, for the most recent
 * forward, the summed duration and the launch count per kernel class: [0] node kernel, [1..4] edge kernel with nn = 8, 16, 32, 64
 * (bench.py's per-kernel roofline). The extra events serialise nothing but cost ~1 us each: not for timed throughput runs. */
int pesto_get_kernel_timing(pesto_model* m, double ms_sum[5], int32_t launches[5]);

/* replaces: extract_topology (src/data_encoding.py:87-102) + the index half of collate_batch_features (src/dataset.py:100-109)
 * for a concatenated batch: exact k nearest neighbours per atom WITHIN its structure, ascending distance, entries with
 * D < 1e-2 (self, coincident atoms) last, emitted as 1-based batch-global ids zero-padded to 64 columns. Distances are rounded
 * exactly as torch.norm rounds them (sqrt(fma(z, z, fma(y, y, x * x))), correctly rounded sqrt); two atoms at exactly the same float32
 * distance are ordered by index (the reference's torch.topk leaves that order undefined).
 * X [n_total,3] and ids_out [n_total,64] follow ptr_kind; struct_offsets [n_struct+1] is a HOST array (offsets[0] = 0,
 * offsets[n_struct] = n_total). */
int pesto_knn_collate(pesto_model* m, int64_t n_total, int32_t n_struct, const int32_t* struct_offsets, const float* X, int32_t k,
                      void* ids_out, int32_t ids_kind, int32_t ptr_kind, void* stream);

/* Companion of pesto_knn_collate, no reference counterpart: the reference takes whatever order torch.topk gives two neighbours at exactly
 * the same float32 distance (src/data_encoding.py:98-99); this library orders them by index. The tables agree as SETS per neighbourhood
 * unless such a tie straddles a layer's cut-off (ids_topk[:, :nn], src/model_operations.py:230): then one of two equally distant atoms is
 * inside the first 8 / 16 / 32 / 64 and the choice is as arbitrary in the reference as here - but logits can differ (0.03 observed on one
 * of 132,417 rows of the reference's pdbs_test set). This call reports those rows so that a caller can warn or pass its own table:
 * flags_out[i] bit 0 / 1 / 2 / 3 = a tie of row i straddles the cut after column 8 / 16 / 32 / 64 of `ids` (the [n_total, 64] table of
 * pesto_knn_collate for the same X / struct_offsets / k). One wave per row, one scan over the row's structure. */
int pesto_knn_tie_rows(pesto_model* m, int64_t n_total, int32_t n_struct, const int32_t* struct_offsets, const float* X, int32_t k,
                       const void* ids, int32_t ids_kind, uint8_t* flags_out, int32_t ptr_kind, void* stream);

/* replaces: the caller-side post-op p = sigmoid(z) (apply_model.ipynb:160, interfaceome/apply_model.py:76) and the
 * residue -> atom expansion of encode_bfactor (src/structure.py:208-218) for per-residue predictions.
 * z [R,n_out] -> p_out [R,n_out] (may be NULL) and bfactor_out [n_out,N] channel-major (may be NULL):
 * bfactor_out[c][i] = sigmoid(z[res_of_atom[i]][c]). With device pointers the call is asynchronous on `stream`, so a bulk
 * loop can keep every structure's result on the GPU and copy back once. */
int pesto_postprocess(pesto_model* m, int64_t N, int64_t R, const float* z, const int32_t* res_of_atom, float* p_out, float* bfactor_out,
                      int32_t ptr_kind, void* stream);

/* replaces: the dense residue mask argument M of Model.forward (model/model.py:32; built by encode_structure, src/data_encoding.py:73,
 * one 1 per row) for callers that hold M itself: M float32 [N,R] (0/1) -> res_of_atom_out int32 [N], the column of each row's single
 * member. One pass over M on the GPU (k_mask_to_segments). A row with zero or several members gives res_of_atom_out[i] = -1, an empty
 * residue column makes res_of_atom_out[0] = -1: the forward that consumes the array then fails its residue-column check
 * (PESTO_ERR_INVALID where the call synchronises, NaN logits otherwise) - SURVEY 8b's argument contract. With device pointers the call
 * is asynchronous on `stream`; with host pointers it returns PESTO_ERR_INVALID itself. */
int pesto_mask_to_segments(pesto_model* m, int64_t N, int64_t R, const float* M, int32_t* res_of_atom_out, int32_t ptr_kind, void* stream);

/* ---- test hooks ----
 * Debug twins of the shipped kernels, selected per handle (the parity tests run every stage through each of them):
 * layer_kernels 0 = shipped (hybrid first layer; arithmetic per the precision policy), 1 = reference-formulation fp32 VALU
 * kernel (LDS-tiled, no MFMA);
 * knn_brute_force != 0: pesto_knn_collate searches every structure by brute force instead of the cell grid. */
int pesto_debug_select(pesto_model* m, int32_t layer_kernels, int32_t knn_brute_force);
/* work decomposition of the shipped state-update kernel: 0 = chosen per launch (default), 1 = rendezvous mode (every wave of a workgroup
 * processes centres, the finish / prepare phase runs behind workgroup rendezvous), 2 = node-wave mode (four waves of twelve only
 * finish / prepare). Both run the same arithmetic in the same order: results must not depend on the choice (tests force each). */
int pesto_debug_edge_mode(pesto_model* m, int32_t mode);

/* ---- per-stage entry points (HOST pointers), used by tests/ to pin each stage against the oracle ----
 * replaces: em.forward (model/model.py:34) */
int pesto_stage_embed(pesto_model* m, int64_t N, const float* q0, float* q_out /*[N,32]*/);
/* replaces: unpack_state_features (src/model_operations.py:6-22); outputs include the sink row 0 */
int pesto_stage_unpack(pesto_model* m, int64_t N, int32_t k, const float* X, const void* ids_topk, int32_t ids_kind,
                       float* D_out /*[N+1,k]*/, float* R_out /*[N+1,k,3]*/);
/* replaces: StateUpdateLayer.forward (src/model_operations.py:225-242) for layer `layer`, using the geometry
 * left by the last pesto_stage_unpack; q [N+1,32] and p [N+1,3,32] updated in place */
int pesto_stage_layer(pesto_model* m, int32_t layer, float* q_io, float* p_io);
/* replaces: StatePoolLayer.forward + decoder (src/model_operations.py:197-213, model/model.py:46-50);
 * q [N,32], p [N,3,32] WITHOUT the sink row */
int pesto_stage_pool(pesto_model* m, int64_t N, int64_t R, const float* q, const float* p, const int32_t* res_of_atom,
                     float* qr_out /*[R,32]*/, float* pr_out /*[R,3,32]*/, float* z_out /*[R,n_out]*/);

#ifdef __cplusplus
}
#endif
#endif /* PESTO_HIP_H */
