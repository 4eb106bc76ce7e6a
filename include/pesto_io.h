/* pesto_io.h - C ABI of libpesto_io.so: the structure I/O either side of the forward pass, in native code (host only,
 * no HIP: safe to use from forked data-loader workers). SURVEY.md 8(f) row 2.
 *
 * The reference does this work in Python on dicts of numpy arrays, on top of the third-party gemmi parser:
 *   read_pdb                         src/structure_io.py:6-55   (gemmi.read_pdb(path, max_line_length=80), all models)
 *   clean_structure                  src/structure.py:14-56
 *   tag_hetatm_chains                src/structure.py:95-110
 *   split_by_chain                   src/structure.py:63-80
 *   filter_non_atomic_subunits       src/structure.py:137-146
 *   remove_duplicate_tagged_subunits src/structure.py:113-134
 *   concatenate_chains               src/structure.py:83-92
 *   encode_structure/encode_features src/data_encoding.py:61-84
 *   encode_bfactor + save_pdb        src/structure.py:185-223, src/structure_io.py:96-123
 * chained as StructuresDataset.__getitem__ (src/dataset.py:126-156) and apply_model.ipynb cell 6.
 * A pesto_structure is that dict: one entry per atom of xyz, name, element, resname, resid, het_flag, chain_name, icode.
 * All functions return 0 or a negative code; pesto_io_last_error() holds the thread-local message.
 */
#ifndef PESTO_IO_H
#define PESTO_IO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pesto_structure pesto_structure;

enum { PESTO_IO_OK = 0, PESTO_IO_ERR_INVALID = -1, PESTO_IO_ERR_FILE = -2, PESTO_IO_ERR_PARSE = -3 };

/* text fields of pesto_io_get_text / pesto_io_from_arrays */
enum { PESTO_IO_NAME = 0, PESTO_IO_ELEMENT = 1, PESTO_IO_RESNAME = 2, PESTO_IO_HET_FLAG = 3, PESTO_IO_CHAIN_NAME = 4, PESTO_IO_ICODE = 5 };

/* steps of pesto_io_preprocess, applied in this order (StructuresDataset.__getitem__, src/dataset.py:139-153) */
enum {
    PESTO_IO_CLEAN = 1,           /* clean_structure: drop water / H / D, renumber residues 1.. over (chain, resid, icode) changes, drop icode */
    PESTO_IO_TAG_HETATM = 2,      /* tag_hetatm_chains: every HETATM residue becomes its own chain "<chain>:<counter>" (names cut to 10 chars) */
    PESTO_IO_SPLIT = 4,           /* split_by_chain + concatenate_chains: atoms grouped by chain name in sorted-name order */
    PESTO_IO_FILTER_NON_ATOMIC = 8,   /* filter_non_atomic_subunits: drop chains with one atom per residue and more than one atom */
    PESTO_IO_REMOVE_DUPLICATES = 16,  /* remove_duplicate_tagged_subunits: drop a tagged chain overlapping (< 0.2 A) an earlier one of equal size */
    PESTO_IO_ALL = 31
};

const char* pesto_io_last_error(void);

/* replaces: read_pdb(pdb_filepath) (src/structure_io.py:6-55): ATOM/HETATM records of every MODEL, lines cut at 80 columns,
 * first-seen alternate location per (chain, residue number, atom name), chain_name = "<chain>:<model index>". */
int pesto_io_read_pdb(const char* path, pesto_structure** out);
int pesto_io_parse_pdb(const char* text, int64_t len, pesto_structure** out);

/* builds a structure from caller arrays (the reference's dict): text fields are fixed-width, NUL-padded char arrays of
 * `width[f]` bytes per atom indexed by the PESTO_IO_* field ids; icode may be NULL (already cleaned). */
int pesto_io_from_arrays(int64_t n, const float* xyz, const int64_t* resid, const char* const text[6], const int32_t width[6],
                         pesto_structure** out);

int pesto_io_free(pesto_structure* s);

/* replaces: the preprocessing chain of StructuresDataset.__getitem__ (+ concatenate_chains); `steps` = OR of PESTO_IO_* */
int pesto_io_preprocess(pesto_structure* s, int32_t steps);

int pesto_io_n_atoms(const pesto_structure* s, int64_t* n);
int pesto_io_get_xyz(const pesto_structure* s, float* xyz /*[N,3]*/);
int pesto_io_get_resid(const pesto_structure* s, int64_t* resid /*[N]*/);
/* copies a text field as fixed-width NUL-padded records of `width` bytes (longer values are an error) */
int pesto_io_get_text(const pesto_structure* s, int32_t field, char* out, int32_t width);

/* replaces: encode_structure + encode_features (src/data_encoding.py:61-84) for the model input:
 * X [N,3]; q0 [N,n0] one-hot with an "unknown" last column per block - n0 = 30: elements (i_v4_*), n0 = 123:
 * elements | residue names | atom names (i_v3_*); res_of_atom [N] = column of the reference's mask M (rank of the atom's
 * resid among the sorted unique resids); n_res = number of columns. Any output pointer may be NULL. */
int pesto_io_encode(const pesto_structure* s, int32_t n0, float* X, float* q0, int32_t* res_of_atom, int64_t* n_res);

/* replaces: encode_bfactor (per-atom or per-residue p, src/structure.py:185-223) + split_by_chain + save_pdb
 * (src/structure_io.py:96-123). bfactor: NULL (0.0), n_values == N (per atom) or n_values == n_res (per residue, expanded).
 * pesto_io_format_pdb returns the text in a buffer owned by the structure (valid until the next call on it). */
int pesto_io_write_pdb(const pesto_structure* s, const float* bfactor, int64_t n_values, const char* path);
int pesto_io_format_pdb(pesto_structure* s, const float* bfactor, int64_t n_values, const char** text, int64_t* len);

/* replaces: what Model.forward needs from the dense residue mask M [N,R] of encode_structure (src/data_encoding.py:73; used by
 * StatePoolLayer, src/model_operations.py:199): res_of_atom[i] = the one column with M[i][r] > 0.5. PESTO_IO_ERR_INVALID for a row
 * with zero or several members or an empty column (the reference's dense softmax degenerates there). Host arrays; the mask on the
 * GPU is reduced by pesto_mask_to_segments (pesto_hip.h). */
int pesto_io_mask_to_segments(const float* M, int64_t N, int64_t R, int32_t* res_of_atom);
/* the same for a mask of 1-byte elements - what encode_structure itself returns (numpy bool, src/data_encoding.py:61-75; callers only
 * turn it into floats for torch, apply_model.ipynb:155) - or 4-byte floats (elem_bytes 1 | 4). Every row is checked in full (byte / word
 * sums, AVX2 where the host has it); the member's column is looked for at the previous row's column and the one after it first (the
 * reference's columns follow the contiguous residue numbering of clean_structure, src/structure.py:47), anywhere else otherwise. */
int pesto_io_mask_to_segments_any(const void* M, int32_t elem_bytes, int64_t N, int64_t R, int32_t* res_of_atom);

#ifdef __cplusplus
}
#endif
#endif /* PESTO_IO_H */
