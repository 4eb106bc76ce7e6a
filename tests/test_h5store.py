"""HDF5 result store (pesto_amd/h5store.py): the reference's ``hf[key] = p.cpu().numpy()`` (interfaceome/apply_model.py:53-79) through the
HDF5 C library, no h5py. CPU tests; skipped as a whole only where the machine has no libhdf5 (the ROCm image has /opt/conda/lib/libhdf5)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from pesto_amd import h5store

pytestmark = pytest.mark.skipif(not h5store.available(), reason="no HDF5 C library on this machine")

H5DUMP = shutil.which("h5dump") or ("/opt/conda/bin/h5dump" if os.path.exists("/opt/conda/bin/h5dump") else None)


def _tables():
    rng = np.random.default_rng(11)
    # keys of the reference's stores: pdbid / assembly / chain:index (src/dataset.py:10), and a plain name
    return {"1ABC/1/A:0": rng.uniform(0, 1, (122, 5)).astype(np.float32), "1ABC/1/B:0": rng.uniform(0, 1, (7, 5)).astype(np.float32),
            "2XYZ/2/C:1": rng.uniform(0, 1, (1, 5)).astype(np.float32), "single": rng.uniform(0, 1, (40, 1)).astype(np.float32)}


def test_store_round_trip_groups_dtypes_and_errors(tmp_path):
    path = str(tmp_path / "out.h5")
    tabs = _tables()
    with h5store.H5Store(path, "w") as hf:
        for k, v in tabs.items():
            hf[k] = v
        hf["ints/i64"] = np.arange(-3, 4, dtype=np.int64)
        hf["ints/u8"] = np.array([[True, False], [False, True]])                # bool -> uint8, as h5py stores numpy bool (enum there; bytes here)
        hf["f64"] = np.linspace(0, 1, 9).reshape(3, 3)
        hf["empty"] = np.zeros((0, 5), np.float32)
        hf["noncontiguous"] = np.arange(12, dtype=np.float32).reshape(3, 4).T
        with pytest.raises(h5store.H5Error, match="already exists"):            # h5py raises on a second hf[key] = ... too
            hf["single"] = tabs["single"]
        with pytest.raises(TypeError):
            hf["c"] = np.zeros(3, np.complex64)
        with pytest.raises(KeyError):
            hf["/"] = np.zeros(3)
    assert open(path, "rb").read(8) == b"\x89HDF\r\n\x1a\n"
    with h5store.H5Store(path) as hf:
        assert hf.keys() == sorted(list(tabs) + ["ints/i64", "ints/u8", "f64", "empty", "noncontiguous"])
        for k, v in tabs.items():
            got = hf[k]
            assert got.dtype == np.float32 and np.array_equal(got, v)
        assert np.array_equal(hf["ints/i64"], np.arange(-3, 4)) and hf["ints/i64"].dtype == np.int64
        assert hf["ints/u8"].dtype == np.uint8 and hf["ints/u8"].tolist() == [[1, 0], [0, 1]]
        assert hf["f64"].dtype == np.float64 and hf["empty"].shape == (0, 5)
        assert np.array_equal(hf["noncontiguous"], np.arange(12, dtype=np.float32).reshape(3, 4).T)
        assert "1ABC/1/A:0" in hf and "1ABC/1" in hf and "1ABC/3/A:0" not in hf and "nope/x" not in hf
        with pytest.raises(KeyError):
            hf["1ABC/1/Z:9"]
        with pytest.raises(KeyError):
            hf["1ABC/1"]                                                         # a group is not a dataset
        with pytest.raises(h5store.H5Error, match="read-only"):
            hf["new"] = np.zeros(2, np.float32)
    with pytest.raises(h5store.H5Error, match="cannot open"):
        h5store.H5Store(str(tmp_path / "missing.h5"))
    (tmp_path / "text.h5").write_text("not an hdf5 file")
    with pytest.raises(h5store.H5Error, match="cannot open"):
        h5store.H5Store(str(tmp_path / "text.h5"))


@pytest.mark.skipif(H5DUMP is None, reason="no h5dump tool")
def test_the_hdf5_projects_own_reader_sees_what_the_reference_would_have_written(tmp_path):
    """Independent reader: h5dump (HDF5 project) must list one contiguous little-endian float32 dataset of the right shape per key - the
    defaults of h5py's ``hf[key] = float32 array`` - and print the same numbers."""
    path = str(tmp_path / "out.h5")
    tabs = _tables()
    with h5store.H5Store(path, "w") as hf:
        for k, v in tabs.items():
            hf[k] = v
    names = subprocess.run([H5DUMP, "-n", path], check=True, capture_output=True, text=True).stdout
    for k in tabs:
        assert f"dataset    /{k}\n" in names
    assert "group      /1ABC/1\n" in names
    head = subprocess.run([H5DUMP, "-H", "-p", "-d", "/1ABC/1/B:0", path], check=True, capture_output=True, text=True).stdout
    assert "H5T_IEEE_F32LE" in head and "( 7, 5 ) / ( 7, 5 )" in head and "CONTIGUOUS" in head and "".join(head[head.index("FILTERS {"):].split()).startswith("FILTERS{NONE}")
    body = subprocess.run([H5DUMP, "-d", "/1ABC/1/B:0", "-y", "-w", "0", "-m", "%.9g", path], check=True, capture_output=True, text=True).stdout
    data = body[body.index("DATA {") + 6:body.rindex("}")]
    vals = np.array([float(t) for t in data.replace("}", " ").replace(",", " ").split()], dtype=np.float32)
    assert np.array_equal(vals.reshape(7, 5), tabs["1ABC/1/B:0"])


def test_bulk_result_file_as_hdf5(tmp_path):
    """apply.save_results / load_results with an .h5 name: one dataset per structure, like the reference's output store."""
    from pesto_amd.apply import load_results, save_results
    rng = np.random.default_rng(4)
    res = {f"/data/pdb/{k}.pdb": rng.uniform(0, 1, (n, 5)).astype(np.float32) for k, n in (("1abc_A", 122), ("2xyz_B:0", 7), ("q", 1))}
    path = save_results(res, str(tmp_path / "out.h5"))
    assert not os.path.exists(path + ".tmp")
    back = load_results(path)
    assert sorted(back) == sorted(k.lstrip("/") for k in res) and all(np.array_equal(back[k.lstrip("/")], res[k]) for k in res)
    named = save_results(res, str(tmp_path / "named.hdf5"), keys={k: f"X{i}/1/A:0" for i, k in enumerate(res)})
    back = load_results(named)
    assert sorted(back) == ["X0/1/A:0", "X1/1/A:0", "X2/1/A:0"] and np.array_equal(back["X1/1/A:0"], res["/data/pdb/2xyz_B:0.pdb"])
    assert load_results(save_results({}, str(tmp_path / "empty.h5"))) == {}


def test_no_library_is_an_error_not_another_format(tmp_path, monkeypatch):
    """PESTO_HDF5_LIB pointing nowhere: H5Unavailable naming what was tried; save_results writes nothing."""
    from pesto_amd.apply import save_results
    monkeypatch.setattr(h5store, "_lib", None)
    monkeypatch.setenv("PESTO_HDF5_LIB", str(tmp_path / "libhdf5_missing.so"))
    try:
        assert not h5store.available()
        with pytest.raises(h5store.H5Unavailable, match="libhdf5_missing.so"):
            save_results({"a": np.zeros((2, 5), np.float32)}, str(tmp_path / "out.h5"))
        assert not os.path.exists(tmp_path / "out.h5") and not os.path.exists(tmp_path / "out.h5.tmp")
    finally:
        monkeypatch.delenv("PESTO_HDF5_LIB")
        monkeypatch.setattr(h5store, "_lib", None)
        assert h5store.available()


def test_link_visitor_is_resolved_lazily_under_any_of_its_exported_names(tmp_path, monkeypatch):
    """HDF5 >= 1.12 exports H5Lvisit2 / H5Lvisit1 and keeps H5Lvisit as a header macro. A library without the plain name must
    still load (writes never need the visitor) and list keys through whichever name it has; none of the three is an error of keys() only."""
    lib = h5store.load()
    real = h5store._link_visitor(lib)

    class Proxy:      # the loaded library with a chosen set of visitor symbols
        def __init__(self, have):
            self._have, self._name = have, lib._name

        def __getattr__(self, name):
            if name in ("H5Lvisit", "H5Lvisit1", "H5Lvisit2"):
                if name in self._have:
                    return real
                raise AttributeError(name)
            return getattr(lib, name)

    path = str(tmp_path / "v.h5")
    for have in (("H5Lvisit",), ("H5Lvisit2", "H5Lvisit1"), ("H5Lvisit1",)):
        monkeypatch.setattr(h5store, "_lib", Proxy(have))
        with h5store.H5Store(path, "w") as hf:
            hf["g/a"] = np.arange(3, dtype=np.float32)
        with h5store.H5Store(path) as hf:
            assert hf.keys() == ["g/a"]
    monkeypatch.setattr(h5store, "_lib", Proxy(()))
    with h5store.H5Store(path, "w") as hf:      # a write-only use does not depend on the visitor
        hf["b"] = np.zeros(2, np.float32)
    with h5store.H5Store(path) as hf:
        assert np.array_equal(hf["b"], np.zeros(2, np.float32))
        with pytest.raises(h5store.H5Unavailable, match="H5Lvisit2"):
            hf.keys()
    monkeypatch.setattr(h5store, "_lib", lib)


def test_result_path_problems_are_found_before_the_work(tmp_path, monkeypatch):
    """colliding dataset names ('/a.pdb' and 'a.pdb') and a missing HDF5 library are raised by the up-front check apply_model
    makes (check_results_path), not in the middle of the final write; a failed write leaves no temporary file behind."""
    from pesto_amd.apply import check_results_path, h5_dataset_names, save_results
    tab = np.zeros((2, 5), np.float32)
    with pytest.raises(ValueError, match="same HDF5 dataset name"):
        check_results_path(str(tmp_path / "o.h5"), ["/a.pdb", "a.pdb"])
    with pytest.raises(ValueError, match="same HDF5 dataset name"):
        save_results({"/a.pdb": tab, "a.pdb": tab}, str(tmp_path / "o.h5"))
    assert not os.path.exists(tmp_path / "o.h5") and not os.path.exists(tmp_path / "o.h5.tmp")
    assert h5_dataset_names(["/x/a.pdb", "b"], {"b": "B/1"}) == {"/x/a.pdb": "x/a.pdb", "b": "B/1"}
    check_results_path(str(tmp_path / "o.npz"), ["/a.pdb", "a.pdb"])      # the .npz form keeps the keys as they are
    check_results_path(None, ["/a.pdb", "a.pdb"])
    # a write that fails half way (a dataset name that is also a group of another) removes its temporary file
    with pytest.raises(h5store.H5Error):
        save_results({"a": tab, "b": tab}, str(tmp_path / "p.h5"), keys={"a": "g", "b": "g/x"})
    assert not os.path.exists(tmp_path / "p.h5") and not os.path.exists(tmp_path / "p.h5.tmp")
    monkeypatch.setattr(h5store, "_lib", None)
    monkeypatch.setenv("PESTO_HDF5_LIB", str(tmp_path / "libhdf5_missing.so"))
    try:
        with pytest.raises(h5store.H5Unavailable):
            check_results_path(str(tmp_path / "q.h5"), ["a"])
    finally:
        monkeypatch.delenv("PESTO_HDF5_LIB")
        monkeypatch.setattr(h5store, "_lib", None)
        assert h5store.available()
