"""Host mirror of the reference's structure I/O over libpesto_io.so (include/pesto_io.h) - SURVEY 8f row 2.

Same names and dict layout as the reference so a caller can switch imports:
    read_pdb                      src/structure_io.py:6-55   (the reference needs gemmi; this does not)
    StructuresDataset             src/dataset.py:114-156
    clean_structure ... concatenate_chains   src/structure.py:14-146   (here: Structure.preprocess(steps))
    encode_structure / encode_features       src/data_encoding.py:61-84 (here: Structure.encode(n0))
    encode_bfactor + save_pdb     src/structure.py:185-223, src/structure_io.py:96-123

A ``structure`` is the reference's dict of per-atom numpy arrays: xyz float32 [N,3], name, element, resname, het_flag,
chain_name (str arrays), resid (int), icode (until cleaned). All work happens in native code; there is no Python fallback.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PESTO_IO_LIB", os.path.join(_HERE, "csrc", "libpesto_io.so"))

NAME, ELEMENT, RESNAME, HET_FLAG, CHAIN_NAME, ICODE = range(6)
CLEAN, TAG_HETATM, SPLIT, FILTER_NON_ATOMIC, REMOVE_DUPLICATES, ALL = 1, 2, 4, 8, 16, 31
_FIELDS = {"name": NAME, "element": ELEMENT, "resname": RESNAME, "het_flag": HET_FLAG, "chain_name": CHAIN_NAME, "icode": ICODE}
_WIDTH = 16

# every symbol include/pesto_io.h declares (tests/test_abi.py)
ABI_SYMBOLS = [
    "pesto_io_last_error", "pesto_io_read_pdb", "pesto_io_parse_pdb", "pesto_io_from_arrays", "pesto_io_free",
    "pesto_io_preprocess", "pesto_io_n_atoms", "pesto_io_get_xyz", "pesto_io_get_resid", "pesto_io_get_text",
    "pesto_io_encode", "pesto_io_write_pdb", "pesto_io_format_pdb", "pesto_io_mask_to_segments", "pesto_io_mask_to_segments_any",
]

_lib = None


class PestoIOError(RuntimeError):
    pass


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PestoIOError(f"{LIB_PATH} not found: build it with `python -m pesto_amd.csrc.build`")
    lib = ctypes.CDLL(LIB_PATH)
    c_p, i32, i64, P = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.POINTER
    lib.pesto_io_last_error.restype = ctypes.c_char_p
    lib.pesto_io_last_error.argtypes = []
    lib.pesto_io_read_pdb.argtypes = [ctypes.c_char_p, P(c_p)]
    lib.pesto_io_parse_pdb.argtypes = [ctypes.c_char_p, i64, P(c_p)]
    lib.pesto_io_from_arrays.argtypes = [i64, c_p, c_p, P(c_p), P(i32), P(c_p)]
    lib.pesto_io_free.argtypes = [c_p]
    lib.pesto_io_preprocess.argtypes = [c_p, i32]
    lib.pesto_io_n_atoms.argtypes = [c_p, P(i64)]
    lib.pesto_io_get_xyz.argtypes = [c_p, c_p]
    lib.pesto_io_get_resid.argtypes = [c_p, c_p]
    lib.pesto_io_get_text.argtypes = [c_p, i32, c_p, i32]
    lib.pesto_io_encode.argtypes = [c_p, i32, c_p, c_p, c_p, P(i64)]
    lib.pesto_io_write_pdb.argtypes = [c_p, c_p, i64, ctypes.c_char_p]
    lib.pesto_io_format_pdb.argtypes = [c_p, c_p, i64, P(ctypes.c_char_p), P(i64)]
    lib.pesto_io_mask_to_segments.argtypes = [c_p, i64, i64, c_p]
    lib.pesto_io_mask_to_segments_any.argtypes = [c_p, ctypes.c_int32, i64, i64, c_p]
    for name in ABI_SYMBOLS:
        if name != "pesto_io_last_error":
            getattr(lib, name).restype = ctypes.c_int
    _lib = lib
    return lib


def _check(rc):
    if rc != 0:
        msg = load().pesto_io_last_error()
        raise PestoIOError(f"libpesto_io error {rc}: {msg.decode() if msg else '?'}")


class Structure:
    """Owner of a native pesto_structure handle."""

    def __init__(self, handle):
        self._h = handle

    def __del__(self):
        try:
            if self._h:
                load().pesto_io_free(self._h)
                self._h = None
        except Exception:
            pass

    # ---- construction
    @classmethod
    def read_pdb(cls, path):
        h = ctypes.c_void_p()
        _check(load().pesto_io_read_pdb(os.fsencode(path), ctypes.byref(h)))
        return cls(h)

    @classmethod
    def parse_pdb(cls, text):
        data = text.encode() if isinstance(text, str) else bytes(text)
        h = ctypes.c_void_p()
        _check(load().pesto_io_parse_pdb(data, len(data), ctypes.byref(h)))
        return cls(h)

    @classmethod
    def from_dict(cls, structure):
        """From the reference's dict (str or bytes arrays; 'icode' optional, 'chain_name' optional -> blank)."""
        n = int(np.asarray(structure["xyz"]).shape[0])
        xyz = np.ascontiguousarray(structure["xyz"], dtype=np.float32)
        resid = np.ascontiguousarray(structure["resid"], dtype=np.int64)
        bufs, ptrs, widths = [], (ctypes.c_void_p * 6)(), (ctypes.c_int32 * 6)()
        for key, f in _FIELDS.items():
            if key not in structure:
                if key == "icode":
                    bufs.append(None); ptrs[f] = None; widths[f] = 0
                    continue
                if key != "chain_name":
                    raise KeyError(key)
                arr = np.zeros(n, dtype="S1")
            else:
                arr = np.char.encode(np.asarray(structure[key]).astype(str), "ascii") if n else np.zeros(0, dtype="S1")
            w = max(1, arr.dtype.itemsize)
            arr = np.ascontiguousarray(arr.astype(f"S{w}"))
            bufs.append(arr); ptrs[f] = arr.ctypes.data; widths[f] = w
        h = ctypes.c_void_p()
        _check(load().pesto_io_from_arrays(n, xyz.ctypes.data, resid.ctypes.data, ptrs, widths, ctypes.byref(h)))
        return cls(h)

    # ---- the reference's preprocessing chain
    def preprocess(self, steps=ALL):
        _check(load().pesto_io_preprocess(self._h, steps))
        return self

    # ---- access
    def __len__(self):
        n = ctypes.c_int64()
        _check(load().pesto_io_n_atoms(self._h, ctypes.byref(n)))
        return n.value

    def _text(self, field):
        buf = np.zeros(len(self), dtype=f"S{_WIDTH}")
        _check(load().pesto_io_get_text(self._h, field, buf.ctypes.data, _WIDTH))
        return np.char.decode(buf, "ascii") if len(buf) else buf.astype(str)

    def to_dict(self):
        n = len(self)
        xyz = np.empty((n, 3), np.float32)
        resid = np.empty(n, np.int64)
        _check(load().pesto_io_get_xyz(self._h, xyz.ctypes.data))
        _check(load().pesto_io_get_resid(self._h, resid.ctypes.data))
        d = {"xyz": xyz, "name": self._text(NAME), "element": self._text(ELEMENT), "resname": self._text(RESNAME), "resid": resid,
             "het_flag": self._text(HET_FLAG), "chain_name": self._text(CHAIN_NAME)}
        try:
            d["icode"] = self._text(ICODE)
        except PestoIOError:
            pass
        return d

    def subunits(self):
        """split_by_chain (src/structure.py:63-80): {chain_name: structure dict without 'chain_name'} in sorted-name order."""
        d = self.to_dict()
        cn = d.pop("chain_name")
        return {c: {k: v[cn == c] for k, v in d.items()} for c in np.unique(cn)}

    # ---- model inputs
    def encode(self, n0=30):
        """(X [N,3] f32, q0 [N,n0] f32 one-hot, res_of_atom [N] int32, R): encode_structure + encode_features with M given as its
        column index per atom (M[i, res_of_atom[i]] = 1)."""
        n = len(self)
        X = np.empty((n, 3), np.float32)
        q0 = np.empty((n, n0), np.float32)
        roa = np.empty(n, np.int32)
        R = ctypes.c_int64()
        _check(load().pesto_io_encode(self._h, n0, X.ctypes.data, q0.ctypes.data, roa.ctypes.data, ctypes.byref(R)))
        return X, q0, roa, R.value

    def mask(self):
        """The reference's dense boolean M [N, R] (src/data_encoding.py:73)."""
        _, _, roa, R = self.encode(30)
        M = np.zeros((len(self), R), dtype=bool)
        M[np.arange(len(self)), roa] = True
        return M

    # ---- output
    def _bf(self, bfactor):
        if bfactor is None:
            return None, 0
        bf = np.ascontiguousarray(bfactor.detach().cpu().numpy() if hasattr(bfactor, "detach") else bfactor, dtype=np.float32).ravel()
        return bf, bf.size

    def save_pdb(self, path, bfactor=None):
        """encode_bfactor (per atom or per residue) + save_pdb(split_by_chain(structure), path)."""
        bf, n = self._bf(bfactor)
        _check(load().pesto_io_write_pdb(self._h, bf.ctypes.data if bf is not None else None, n, os.fsencode(path)))

    def format_pdb(self, bfactor=None):
        bf, n = self._bf(bfactor)
        t, ln = ctypes.c_char_p(), ctypes.c_int64()
        _check(load().pesto_io_format_pdb(self._h, bf.ctypes.data if bf is not None else None, n, ctypes.byref(t), ctypes.byref(ln)))
        return ctypes.string_at(t, ln.value).decode()


# ------------------------------------------------------------------ reference-named entry points
def read_pdb(pdb_filepath):
    """src/structure_io.py:6-55 -> the same dict of arrays (xyz, name, element, resname, resid, het_flag, chain_name, icode)."""
    d = Structure.read_pdb(pdb_filepath).to_dict()
    d["resid"] = d["resid"].astype(np.int32)
    return d


def save_pdb(subunits, filepath):
    """src/structure_io.py:96-123: subunits = {chain_name: structure dict (optionally with 'bfactor')}."""
    parts, bf = [], []
    for cn, su in subunits.items():
        d = {k: v for k, v in su.items() if k != "bfactor"}
        d["chain_name"] = np.array([cn] * np.asarray(su["xyz"]).shape[0])
        parts.append(d)
        if "bfactor" in su:
            bf.append(np.asarray(su["bfactor"], dtype=np.float32))
    keys = set.intersection(*[set(p) for p in parts])
    st = {k: np.concatenate([np.asarray(p[k]) for p in parts]) for k in keys}
    Structure.from_dict(st).save_pdb(filepath, np.concatenate(bf) if len(bf) == len(parts) else None)


class StructuresDataset:
    """src/dataset.py:114-156: item i = (subunits, path) after the reference's preprocessing chain, or (None, path) when
    the file cannot be read. ``as_structure=True`` returns the native Structure (preprocessed and concatenated) instead
    of the dict of subunits - the form Structure.encode / Model consume directly."""

    def __init__(self, pdb_filepaths, with_preprocessing=True, as_structure=False):
        self.pdb_filepaths = pdb_filepaths
        self.with_preprocessing = with_preprocessing
        self.as_structure = as_structure

    def __len__(self):
        return len(self.pdb_filepaths)

    def __getitem__(self, i):
        path = self.pdb_filepaths[i]
        try:
            s = Structure.read_pdb(path)
        except PestoIOError as e:
            print(f"ReadError: {path}: {e}")
            return None, path
        if self.with_preprocessing:
            s.preprocess(ALL)
            return (s if self.as_structure else s.subunits()), path
        return (s if self.as_structure else s.to_dict()), path


# ---------------------------------------------------------------------------------------------- chemical-component CIF
def _cif_tokens(text):
    """CIF 1.1 tokens of a data block: whitespace-separated words, 'single' / "double" quoted strings and ;-delimited text
    fields; # comments are dropped. Quotes end only before whitespace (a ' inside a word, as in atom names like O5', is data)."""
    out, lines, i = [], text.split("\n"), 0
    while i < len(lines):
        line = lines[i]
        if line.startswith(";"):                       # text field up to the next line that starts with ';'
            buf = [line[1:]]
            i += 1
            while i < len(lines) and not lines[i].startswith(";"):
                buf.append(lines[i])
                i += 1
            out.append(("v", "\n".join(buf).strip()))
            i += 1
            continue
        j, n = 0, len(line)
        while j < n:
            c = line[j]
            if c in " \t\r":
                j += 1
            elif c == "#":
                break
            elif c in "'\"":
                k = j + 1
                while k < n and not (line[k] == c and (k + 1 == n or line[k + 1] in " \t\r")):
                    k += 1
                out.append(("v", line[j + 1:k]))
                j = k + 1
            else:
                k = j
                while k < n and line[k] not in " \t\r":
                    k += 1
                out.append(("w", line[j:k]))
                j = k
        i += 1
    return out


def _cif_first_block(text):
    """({tag: value}, {tag: [column values]}) of the FIRST data block: single tag-value pairs and loop_ columns."""
    toks = _cif_tokens(text)
    values, loops, i, seen_block = {}, {}, 0, False
    while i < len(toks):
        kind, t = toks[i]
        low = t.lower() if kind == "w" else ""
        if low.startswith("data_"):
            if seen_block:
                break
            seen_block = True
            i += 1
        elif low == "loop_":
            i += 1
            tags = []
            while i < len(toks) and toks[i][0] == "w" and toks[i][1].startswith("_"):
                tags.append(toks[i][1])
                i += 1
            rows = []
            while i < len(toks) and not (toks[i][0] == "w" and (toks[i][1].startswith("_") or toks[i][1].lower() == "loop_"
                                                                  or toks[i][1].lower().startswith("data_"))):
                rows.append(toks[i][1])
                i += 1
            if tags:
                for c, tag in enumerate(tags):
                    loops[tag] = rows[c::len(tags)]
        elif kind == "w" and t.startswith("_"):
            if i + 1 < len(toks):
                values[t] = toks[i + 1][1]
            i += 2
        else:
            i += 1
    return values, loops


def read_molecule_cif(filepath):
    """(mol, molid) of a chemical-component dictionary entry (ligand libraries) - what the reference's read_molecule_cif returns
    (src/structure_io.py:58-93; there the parsing is gemmi's cif.read_file, absent from this image: its find_value / find_loop
    behaviour on the first data block is restated here): molid = _chem_comp.id; xyz from _chem_comp_atom.model_Cartn_{x,y,z},
    or - when any of those is '?' - from the pdbx_model_Cartn_*_ideal columns; element = _chem_comp_atom.type_symbol. An entry
    whose atom table is a single tag-value set instead of a loop (one atom, e.g. an ion) gives xyz = zeros [1,3] and the element in
    Title case, exactly like the reference."""
    with open(filepath, "r") as fs:
        values, loops = _cif_first_block(fs.read())
    molid = values.get("_chem_comp.id")
    cols = [loops.get(f"_chem_comp_atom.model_Cartn_{a}", []) for a in "xyz"]
    if any("?" in c for c in cols):
        cols = [loops.get(f"_chem_comp_atom.pdbx_model_Cartn_{a}_ideal", []) for a in "xyz"]
    if len(cols[0]) == 0:
        mol = {"xyz": np.zeros((1, 3)), "element": np.array([values.get("_chem_comp_atom.type_symbol", "").lower().title()])}
    else:
        mol = {"xyz": np.array(cols).T.astype(float), "element": np.array(loops.get("_chem_comp_atom.type_symbol", []))}
    return mol, molid
