"""Native structure I/O (libpesto_io.so) against the reference. No GPU needed.
 * the reader + the whole preprocessing chain + the writer: the reference's own examples/*.pdb -> *_i0.pdb pairs (data files
   copied to tests/golden/pdb/): every column of every line except the b-factor value (the i_v4_1 checkpoint that produced it
   is not in the reference tree);
 * each preprocessing stage, the encoders and the b-factor writer: outputs of the reference's Python functions captured by
   tests/golden/make_golden.py --io."""
import gzip
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, golden
from pesto_amd import structure_io as sio
from pesto_amd.structure_io import Structure

PAIRS = ["7KHT_lipid", "1thf_D", "6I9F"]
CASES = ["synthetic", "7KHT_lipid", "1thf_D", "6I9F", "1ZNS_ion"]


def _gz(name):
    return gzip.open(os.path.join(GOLDEN, "pdb", name + ".gz"), "rt").read()


def _text(g, key):
    return g[key].tobytes().decode()


def _s(arr):
    return np.char.decode(arr, "ascii") if arr.dtype.kind == "S" else arr


def test_header_symbols_exported():
    hdr = open(os.path.join(ROOT, "include", "pesto_io.h")).read()
    declared = set(re.findall(r"\b(pesto_io_[a-z_]+)\s*\(", hdr))
    assert declared == set(sio.ABI_SYMBOLS)
    lib = sio.load()
    for name in declared:
        assert hasattr(lib, name), name


@pytest.mark.parametrize("name", PAIRS)
def test_read_preprocess_write_matches_reference_example_outputs(name, tmp_path):
    """apply_model.ipynb cell 6 end to end (minus the model): read_pdb -> clean -> tag -> split -> filter -> dedup ->
    concatenate -> save_pdb equals the file the reference saved, in every column but the b-factor value."""
    src = tmp_path / (name + ".pdb")
    src.write_text(_gz(name + ".pdb"))
    out = tmp_path / "out.pdb"
    Structure.read_pdb(str(src)).preprocess().save_pdb(str(out))
    strip = lambda l: l[:54] + l[66:] if l.startswith(("ATOM", "HETATM")) else l
    got = [strip(l) for l in out.read_text().split("\n")]
    want = [strip(l) for l in _gz(name + "_i0.pdb").split("\n")]
    assert got == want


@pytest.mark.parametrize("name", CASES)
def test_preprocessing_stages_match_reference_functions(name):
    g = golden("io_" + name)
    text = _text(g, "pdb_text")
    s = Structure.parse_pdb(text).preprocess(sio.CLEAN)                       # clean_structure
    d = s.to_dict()
    assert "icode" not in d
    assert np.array_equal(d["resid"], g["clean_resid"])
    assert np.array_equal(d["chain_name"], _s(g["clean_chain"])) and np.array_equal(d["name"], _s(g["clean_name"]))
    s.preprocess(sio.TAG_HETATM)                                               # tag_hetatm_chains
    assert np.array_equal(s.to_dict()["chain_name"], _s(g["tag_chain"]))
    assert list(s.subunits()) == list(_s(g["split_keys"]))                     # split_by_chain (sorted keys)
    f = Structure.parse_pdb(text).preprocess(sio.CLEAN | sio.TAG_HETATM | sio.FILTER_NON_ATOMIC)
    assert list(f.subunits()) == list(_s(g["filter_keys"]))                    # filter_non_atomic_subunits
    s = Structure.parse_pdb(text).preprocess()                                 # ... remove_duplicate_tagged_subunits, concatenate_chains
    assert list(s.subunits()) == list(_s(g["dedup_keys"]))
    d = s.to_dict()
    assert np.array_equal(d["xyz"], g["final_xyz"]) and np.array_equal(d["resid"], g["final_resid"])
    for k in ("name", "element", "resname", "het_flag", "chain_name"):
        assert np.array_equal(d[k], _s(g["final_" + k])), k


@pytest.mark.parametrize("name", CASES)
def test_encode_and_bfactor_writer_match_reference(name, tmp_path):
    g = golden("io_" + name)
    s = Structure.parse_pdb(_text(g, "pdb_text")).preprocess()
    X, q, roa, R = s.encode(123)                                               # encode_structure + encode_features
    assert R == int(g["n_res"]) and np.array_equal(roa, g["M_col"]) and np.array_equal(X, g["final_xyz"])
    assert np.all(q.sum(1) == 3)
    idx = np.stack([q[:, :30].argmax(1), q[:, 30:59].argmax(1), q[:, 59:].argmax(1)], 1)
    assert np.array_equal(idx, g["q_idx"])
    q30 = s.encode(30)[1]
    assert np.array_equal(q30, q[:, :30])
    M = s.mask()
    assert M.shape == (len(s), R) and np.array_equal(M.argmax(1), roa)
    # encode_bfactor (per-residue p) + save_pdb: byte-identical file
    assert s.format_pdb(g["p_res"]) == _text(g, "saved_text")
    s.save_pdb(str(tmp_path / "o.pdb"), g["bfactor"])                           # per-atom values
    assert (tmp_path / "o.pdb").read_text() == _text(g, "saved_text")
    with pytest.raises(sio.PestoIOError):
        s.format_pdb(np.zeros(R + 1, np.float32))


def test_reference_named_entry_points(tmp_path):
    """read_pdb / StructuresDataset / save_pdb keep the reference's signatures and dict layout."""
    g = golden("io_synthetic")
    path = tmp_path / "syn.pdb"
    path.write_text(_text(g, "pdb_text"))
    st = sio.read_pdb(str(path))
    assert set(st) == {"xyz", "name", "element", "resname", "resid", "het_flag", "chain_name", "icode"}
    assert st["xyz"].dtype == np.float32 and st["resid"].dtype == np.int32
    assert set(np.unique(st["chain_name"])) == {"A:0", "B:0", "C:0", "A:1", "B:1", "C:1"}
    assert "A" in set(st["icode"]) and "H" in set(st["het_flag"]) and {"Se", "Zn", "Fe", "H", "D"} <= set(st["element"])
    # alternate locations: the first seen (chain, number, name) key wins - also across models, as in the reference's loop
    ser = (st["resname"] == "SER") & (st["name"] == "OG")
    assert ser.sum() == 1
    ds = sio.StructuresDataset([str(path), str(tmp_path / "missing.pdb")])
    subunits, p = ds[0]
    assert p == str(path) and list(subunits) == list(_s(g["dedup_keys"]))
    assert ds[1] == (None, str(tmp_path / "missing.pdb"))
    for su in subunits.values():
        su["bfactor"] = np.full(su["xyz"].shape[0], 0.5, np.float32)
    sio.save_pdb(subunits, str(tmp_path / "o.pdb"))
    lines = (tmp_path / "o.pdb").read_text().split("\n")
    assert lines[-1] == "END" and lines[0][60:66] == "  0.50" and sum(l == "TER" for l in lines) == len(subunits)
    native, _ = sio.StructuresDataset([str(path)], as_structure=True)[0]
    assert native.format_pdb(np.full(len(native), 0.5, np.float32)) == "\n".join(lines)


def test_parse_errors_are_reported():
    with pytest.raises(sio.PestoIOError):
        Structure.parse_pdb("ATOM      1  N   ALA A   1      30.837\n")           # too short to hold coordinates
    with pytest.raises(sio.PestoIOError):
        Structure.parse_pdb("REMARK nothing here\nEND\n").preprocess()            # no atoms
    water = "HETATM    1  O   HOH A   1       1.000   2.000   3.000  1.00  0.00           O  \n"
    with pytest.raises(sio.PestoIOError):
        Structure.parse_pdb(water).preprocess()


def test_writer_number_formatting_equals_python_format():
    """The writer formats numbers without printf; every line must equal the reference's Python format string
    (src/structure_io.py:118) for awkward float32 values: exact decimal ties, tiny negatives, -0.0, wide numbers."""
    rng = np.random.default_rng(0)
    special = np.array([0.125, -0.125, 0.375, 2.5e-4, -2.5e-4, -0.0, 0.0, 0.0005, -0.0005, 0.9995, 9.9995, 99.995, 999.9995, -999.9995, 1234.5675,
                        9999.9995, -9999.9995, 12345.678, 0.005, 0.015, 0.025, 0.035, 0.045, 1.005, 2.675, 1e-7, -1e-7, 123456.7], np.float32)
    vals = np.concatenate([special, rng.uniform(-500, 500, 3000).astype(np.float32), (rng.integers(-99999, 99999, 2000) / 1000.0).astype(np.float32),
                           (rng.integers(0, 2000, 1000) / 2000.0).astype(np.float32)])
    n = vals.size
    xyz = np.stack([vals, np.roll(vals, 1), np.roll(vals, 2)], 1).astype(np.float32)
    bf = np.roll(vals, 3).astype(np.float32)
    st = {"xyz": xyz, "name": np.array(["CA"] * n), "element": np.array(["C"] * n), "resname": np.array(["ALA"] * n),
          "resid": np.arange(1, n + 1), "het_flag": np.array(["A"] * n), "chain_name": np.array(["A:0"] * n)}
    text = sio.Structure.from_dict(st).format_pdb(bf).split("\n")
    fmt = "{:<6s}{:>5d} {:<4s} {:>3s} {:1s}{:>4d}    {:8.3f}{:8.3f}{:8.3f}{:6.2f}{:6.2f}          {:<2s}  "
    for i in range(n):
        want = fmt.format("ATOM", i + 1, "CA", "ALA", "A", i + 1, xyz[i, 0], xyz[i, 1], xyz[i, 2], bf[i], bf[i], "C")
        assert text[i] == want, (i, text[i], want)
    assert text[n] == "TER" and text[n + 1] == "END"


def test_read_molecule_cif(tmp_path):
    """Chemical-component CIF reader (reference src/structure_io.py:58-93, whose parser is gemmi): id, model coordinates, the
    ideal-coordinate fallback when a model coordinate is '?', quoted atom names, ;-text fields, and the single-atom (no loop) form."""
    from pesto_amd.structure_io import read_molecule_cif
    head = """data_ATP
#
_chem_comp.id                                    ATP
_chem_comp.name                                  "ADENOSINE-5'-TRIPHOSPHATE"
_chem_comp.formula                               'C10 H16 N5 O13 P3'
_chem_comp.pdbx_synonyms
;two lines of
free text with a _tag.like word
;
#
loop_
_chem_comp_atom.comp_id
_chem_comp_atom.atom_id
_chem_comp_atom.type_symbol
_chem_comp_atom.model_Cartn_x
_chem_comp_atom.model_Cartn_y
_chem_comp_atom.model_Cartn_z
_chem_comp_atom.pdbx_model_Cartn_x_ideal
_chem_comp_atom.pdbx_model_Cartn_y_ideal
_chem_comp_atom.pdbx_model_Cartn_z_ideal
"""
    rows = """ATP PG    P 1.200  -0.226 -6.850  1.162  -0.221  -5.685
ATP "O5'" O 2.000  1.0    -3.5    2.1    1.1     -3.4   # a comment
ATP N1    N -1.5   0.25   7.0     -1.4   0.3     7.1
#
loop_
_chem_comp_bond.comp_id
_chem_comp_bond.atom_id_1
ATP PG
data_SECOND
_chem_comp.id XXX
"""
    p = tmp_path / "atp.cif"
    p.write_text(head + rows)
    mol, molid = read_molecule_cif(str(p))
    assert molid == "ATP" and list(mol["element"]) == ["P", "O", "N"] and mol["xyz"].dtype == np.float64
    assert np.allclose(mol["xyz"], [[1.2, -0.226, -6.85], [2.0, 1.0, -3.5], [-1.5, 0.25, 7.0]])
    p.write_text(head + rows.replace("2.000  1.0    -3.5 ", "?      ?      ?    "))       # missing model coordinates -> ideal ones
    mol, _ = read_molecule_cif(str(p))
    assert np.allclose(mol["xyz"], [[1.162, -0.221, -5.685], [2.1, 1.1, -3.4], [-1.4, 0.3, 7.1]])
    ion = tmp_path / "zn.cif"
    ion.write_text("data_ZN\n_chem_comp.id ZN\n_chem_comp_atom.comp_id ZN\n_chem_comp_atom.atom_id ZN\n_chem_comp_atom.type_symbol ZN\n"
                   "_chem_comp_atom.model_Cartn_x 0.000\n_chem_comp_atom.model_Cartn_y 0.000\n_chem_comp_atom.model_Cartn_z 0.000\n")
    mol, molid = read_molecule_cif(str(ion))
    assert molid == "ZN" and mol["xyz"].shape == (1, 3) and not mol["xyz"].any() and list(mol["element"]) == ["Zn"]


def test_bulk_result_file_round_trip(tmp_path):
    """save_results / load_results: the HDF5-free form of the reference's bulk store hf[key] = p (interfaceome/apply_model.py:53-79)."""
    from pesto_amd.apply import load_results, save_results
    rng = np.random.default_rng(4)
    res = {f"/data/pdb/{k}.pdb": rng.uniform(0, 1, (n, 5)).astype(np.float32) for k, n in (("1abc_A", 122), ("2xyz_B:0", 7), ("q", 1))}
    path = save_results(res, str(tmp_path / "out.npz"))
    back = load_results(path)
    assert list(back) == list(res) and all(np.array_equal(back[k], res[k]) for k in res)
    assert load_results(save_results({}, str(tmp_path / "empty.npz"))) == {}
