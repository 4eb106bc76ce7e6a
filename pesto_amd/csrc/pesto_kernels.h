// pesto_kernels.h - launchers of the gfx950 kernels (definitions in pesto_kernels.hip / pesto_node.hip / pesto_edge.hip)
#pragma once
#include <hip/hip_runtime.h>

#include "pesto_schema.h"

namespace pesto {

// N atoms in the batch; q0 has nq rows used with period nq (nq = N, or the frame length of a trajectory batch)
// p_zero (optional): [N + 1, 96] state array that is zeroed along the way (p0 = zeros), together with the sink row of q_state
// optional extra job of the unpack launch: the residue segment bounds of the pool layer (lo_enc / hi zero-initialised before it)
struct SegBoundsArgs { const int* roa = nullptr; int* lo_enc = nullptr; int* hi = nullptr; int R = 0; int* err_flag = nullptr; };
// optional extra job of the embed launch (the first of a forward): clear up to three arrays of per-forward words (int counts)
struct ClearArgs { int* p0 = nullptr; int n0 = 0; int* p1 = nullptr; int n1 = 0; int* p2 = nullptr; int n2 = 0; };
// range guard of the f16-split kernels, per structure: flags = the launch's flags word (bit 2: some activation left the f16 range),
// sflags = one word per structure of the launch (same bit). Structure of state row i (1-based; 0 = sink): seg_of_atom[i - 1] for a
// ragged batch with separate-call semantics, (i - 1) / frame_n for trajectory frames, 0 for a plain collated call (both null / 0).
// The context lives in device memory SATCTX_OFFSET_INTS ints behind the flags word the layer kernels get (written by the embed launch,
// the first of every forward): the kernels' hot paths carry no extra arguments, only the rare flagging path reads it.
// state_limit: conditioning trigger of PESTO_PRECISION_AUTO (round 5, include/pesto_hip.h::pesto_set_auto_state_limit). The split's dropped
// term is 2^-22 RELATIVE, so its absolute error grows with the states. A finishing wave that sees max |new state| of a centre above the
// limit sets the SAME guard bit as an overflow: the structure is repeated on the exact kernels. +inf (precision f16_split / fp32, or the
// trigger switched off) = never.
// pad_trigger (round 5): a structure with zero-padded neighbour slots (fewer than 64 atoms, or a table of fewer than 64 columns) is flagged by the
// unpack launch like a range overflow and repeated on the exact kernels by PESTO_PRECISION_AUTO: padded slots (wrap-around geometry against
// the sink's zero state, src/model_operations.py:8,17) are where the forward is ill-conditioned for ANY fp32 evaluation - on the pinned
// inputs of tests/golden/fuzz_pins.npz all of the deviation sits in the padded members (the 2-atom member of a collated batch: 1.3e-4
// on its one residue, every other residue <= 2e-5; the reference's own fp32 run is 1.15e-4 off there). Real structures have no such slots.
// only_flagged (round 5): the launch is AUTO's fp32 repeat - the exact layer kernels skip the work items (k_edge) / atom tiles (k_node) whose
// atoms all belong to structures whose guard word is clear: their logits are not rewritten (launch_pool's only_flagged), and a structure's
// neighbours are its own atoms and the sink row, so nothing a flagged structure reads is skipped. One small flagged structure in a
// 24,000-atom launch then costs 32 nearly empty launch pairs instead of the whole launch at fp32 speed.
struct SatCtx { int* flags = nullptr; int* sflags = nullptr; const int* seg_of_atom = nullptr; int frame_n = 0; float state_limit = __builtin_huge_valf();
                int pad_trigger = 0; int only_flagged = 0; };
constexpr int SATCTX_OFFSET_INTS = 3;      // flags buffer: [0] unused, [1] the flags word, [2] collate's copy, [3] pad, [4..] SatCtx (16-byte aligned)
void launch_embed(hipStream_t st, const float* W, const MlpW& em, int N, int nq, int n0, const float* q0, float* q_state, float* p_zero = nullptr,
                  ClearArgs clr = ClearArgs(), SatCtx sc = SatCtx());      // sc.flags non-null: the context is stored behind that word
// F coordinate frames of Nf atoms sharing one ids table [Nf,k] (F = 1: a plain collated batch); X strides in floats;
// dmax_bits[F] must be zeroed. seg_of_atom / seg_end (F = 1 only, may be null): ragged structures that must behave like separate
// calls - per-structure wrap-around target and max(D); dmax_bits then holds one zeroed word per structure
void launch_unpack(hipStream_t st, int Nf, int F, int k, const float* X, int64_t xs_frame, int64_t xs_atom, const void* ids,
                   int ids_kind, int* ids_s, float4* geo, unsigned* dmax_bits, int* err_flag, const int* seg_of_atom = nullptr,
                   const int* seg_end = nullptr, SegBoundsArgs sb = SegBoundsArgs(), bool skip_pass2 = false, int pad_cols = KMAX);
// pad_cols: neighbour columns a layer can gather from (the config's largest nn): zero ids in THEM are padded slots for AUTO's pad trigger
// pass 2 of the geometry (D += max(D) (D < 1e-2), R /= D, sink row; src/model_operations.py:12-20) as extra workgroups of the node launch
// that writes the first layer's records: small launches (one structure) then pay one dependent launch less (round 5). n = N atoms in total.
struct Unpack2Args { int n = 0; int Nf = 0; int* ids_s = nullptr; float4* geo = nullptr; const unsigned* dmax_bits = nullptr; const int* seg_of_atom = nullptr; };
// workgroups of 512 threads the merged form adds for n atoms (0 = do not merge: the node launch's own workgroups + these must fit one wave of 256 CUs)
int unpack2_merge_blocks(int n_atoms, int N1);
void launch_expand_roa(hipStream_t st, int Nf, int R, int F, const int* roa, int* roa_f, int* err_flag);
void launch_layer_v1(hipStream_t st, const float* W, const LayerW& lw, int N1, const int* ids_s, const float4* geo,
                     const float* q_in, const float* p_in, float* q_out, float* p_out);
// MFMA layer (pesto_node.hip, pesto_edge.hip): per-atom node kernel (finish previous layer / prepare records) + edge kernel.
// flags: the flags word (the SatCtx of the launch lies behind it); the f16-split kernels set bit 2 (value 4) of it and of the
// structure's word when an activation left the f16 range (sat_probe)
void launch_node(hipStream_t st, const float* W, const LayerW* finish, const LayerW* prep, int N1, float* q_state, float* p_state,
                 const float* Z, float* rec_nb, float* rec_cen, int variant, int* flags, Unpack2Args u2 = Unpack2Args());
void launch_edge(hipStream_t st, const float* W, const LayerW& lw, int N1, const int* ids_s, const float4* geo,
                 const float* rec_nb, const float* rec_cen, const float* p_state, float* Z, int max_blocks, int variant, int* flags,
                 const float* q_state = nullptr, float* q_out = nullptr, float* p_out = nullptr, const LayerW* next = nullptr,
                 float* rec_nb_out = nullptr, float* rec_cen_out = nullptr, int mode = 0);
// mode (shipped path only): 0 = work decomposition chosen per launch, 1 = rendezvous mode, 2 = node-wave mode (pesto_debug_edge_mode)
// q_out / p_out non-null (variant 0 only): the edge kernel also applies the layer's output MLPs (finish phase) and writes the new
// state there - p_state / q_state (the old state, still gathered by other workgroups) must be different buffers; Z is not used.
// next non-null: it also writes layer `next`'s records of the new state into rec_nb_out / rec_cen_out (prepare phase; buffers
// different from rec_nb / rec_cen, which other workgroups still read)
// k-NN + collate; with use_grid the structures of at least knn_cell_min() atoms are searched through a uniform cell grid
// (buffers: slots n_struct ints = block of the cell arrays per structure or -1, grids n_struct * knn_grid_struct_bytes(), cell_cnt /
// cell_cur n_slots * knn_cells_per_struct() ints, cell_of n_total ints, sorted n_total float4), smaller ones by brute force; identical
// results either way
void launch_knn_collate(hipStream_t st, int n_total, int n_struct, const int* offsets, const float* X, int k, void* ids_out, int ids_kind,
                        int use_grid, const int* slots, void* grids, int* cell_cnt, int* cell_cur, int* cell_of, void* sorted);
// flags[i] bit c set: an exact distance tie of row i straddles the cut after column 8 << c of the table `ids` ([n_total, 64], 1-based)
void launch_knn_ties(hipStream_t st, int n_total, int n_struct, const int* offsets, const float* X, int k, const void* ids, int ids_kind,
                     unsigned char* flags);
size_t knn_grid_struct_bytes();
int knn_cell_min();
int knn_cells_per_struct();
// meta: device array of {int off, roff, n, r, k; long long idoff} per structure (32 bytes each, see k_collate)
// seg_of_atom [n_total] / seg_end [n_struct]: structure index of every atom and the end offset of every structure (for launch_unpack)
void launch_collate(hipStream_t st, int n_total, int n_struct, const void* meta, const void* ids_raw, int ids_kind, const int* roa_raw,
                    int* ids_out, int* roa_out, int* seg_of_atom, int* seg_end, int* err_flag);
// q0 [N, n0] one-hot from n_idx (1..3) byte index columns per atom; offs[c] = first feature of block c
void launch_onehot(hipStream_t st, int N, int n0, int n_idx, const unsigned char* idx, const int* offs, float* q0, int* err_flag);
void launch_segments(hipStream_t st, int n_total, int n_struct, const int* seg_end, int* seg_of_atom);
// dense mask M [N,R] -> roa [N] (column of the single member per row, -1 for rows with != 1 member; roa[0] = -1 if a column is empty);
// seen: R ints of scratch
void launch_mask_to_segments(hipStream_t st, int N, int R, const float* M, int* roa, int* seen, int gen);
void launch_postprocess(hipStream_t st, int N, int R, int n_out, const float* z, const int* roa, float* p_out, float* bf_out, int* err_flag);
// sc (optional): a residue whose structure has its range-guard word set gets NaN logits; only_flagged: ONLY those residues are written
// (the fp32 repeat of PESTO_PRECISION_AUTO: the other structures keep the logits of the split kernels)
void launch_pool(hipStream_t st, const float* W, const ModelW& mw, int n_out, int N, int R, const float* q, const float* p,
                 const int* roa, float* a_tmp, int* lo, int* hi, int* err_flag, float* qr_out, float* pr_out, float* z_out,
                 bool bounds_ready = false,       // bounds_ready: lo / hi already hold the segment bounds (launch_embed with SegBoundsArgs)
                 SatCtx sc = SatCtx(), bool only_flagged = false);

}  // namespace pesto
