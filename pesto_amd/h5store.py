"""HDF5 result store: what the reference's bulk driver writes, ``hf[key] = p.cpu().numpy()`` per structure into
``h5py.File(output_filepath, 'w')`` (interfaceome/apply_model.py:53-79) - one plain (contiguous, unfiltered) dataset per key, the
``/`` in a key creating the groups on the way, exactly as h5py does.

h5py is not a dependency of this package; the HDF5 C library is bound directly with ctypes (the 1.10 / 1.12 / 1.14 C API used here is
the same: H5Fcreate / H5Dcreate2 / H5Dwrite / H5Dread; the link visitor is exported as H5Lvisit by 1.10 and as H5Lvisit2 / H5Lvisit1
from 1.12 on, where H5Lvisit is a header macro - it is resolved lazily, by keys() only, so writing never depends on it. hid_t is
bound as int64: the 1.10+ ABI). The library is looked for in ``$PESTO_HDF5_LIB``, the loader's
search path and the usual prefixes (``/opt/conda/lib`` in the ROCm image). No library -> H5Unavailable with the places that were tried:
the caller (apply.save_results) then has the .npz store, and says so - nothing is written silently in another format.

Files written here open with ``h5py.File(path)`` / ``h5dump``; files written by the reference's loop (h5py defaults: contiguous
little-endian float32) read back with H5Store(path).  Not covered: filtered datasets (the reference's INPUT stores use h5py's own
"lzf" filter, src/dataset.py:63 - a plugin the C library does not ship), attributes, compound types.
"""
import ctypes
import ctypes.util
import glob
import os

import numpy as np

hid_t = ctypes.c_int64
hsize_t = ctypes.c_uint64
herr_t = ctypes.c_int

H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0, 2
H5P_DEFAULT, H5S_ALL = 0, 0
H5_INDEX_NAME, H5_ITER_NATIVE = 0, 2
H5T_INTEGER, H5T_FLOAT = 0, 1
H5T_ORDER_LE = 0
H5T_SGN_NONE = 0


class H5Unavailable(RuntimeError):
    pass


class H5Error(RuntimeError):
    pass


_lib = None


def _candidates():
    env = os.environ.get("PESTO_HDF5_LIB")
    if env:
        yield env
        return                          # an explicit choice is not second-guessed
    found = ctypes.util.find_library("hdf5") or ctypes.util.find_library("hdf5_serial")
    if found:
        yield found
    for prefix in ("/opt/conda/lib", "/usr/lib/x86_64-linux-gnu", "/usr/lib/x86_64-linux-gnu/hdf5/serial", "/usr/local/lib", "/usr/lib64"):
        for name in ("libhdf5.so", "libhdf5_serial.so"):
            for path in sorted(glob.glob(os.path.join(prefix, name + "*"))):
                yield path


def load():
    """The bound library (cached). Raises H5Unavailable when no HDF5 C library can be loaded."""
    global _lib
    if _lib is not None:
        return _lib
    tried = []
    lib = None
    for path in _candidates():
        try:
            lib = ctypes.CDLL(path)
            break
        except OSError as e:
            tried.append(f"{path}: {e}")
    if lib is None:
        raise H5Unavailable("no HDF5 C library could be loaded (set PESTO_HDF5_LIB to a libhdf5.so); tried: " + ("; ".join(tried) or "nothing found"))
    sig = {
        "H5open": (herr_t, []),
        "H5get_libversion": (herr_t, [ctypes.POINTER(ctypes.c_uint)] * 3),
        "H5Eset_auto2": (herr_t, [hid_t, ctypes.c_void_p, ctypes.c_void_p]),
        "H5Fcreate": (hid_t, [ctypes.c_char_p, ctypes.c_uint, hid_t, hid_t]),
        "H5Fopen": (hid_t, [ctypes.c_char_p, ctypes.c_uint, hid_t]),
        "H5Fflush": (herr_t, [hid_t, ctypes.c_int]),
        "H5Fclose": (herr_t, [hid_t]),
        "H5Pcreate": (hid_t, [hid_t]),
        "H5Pset_create_intermediate_group": (herr_t, [hid_t, ctypes.c_uint]),
        "H5Pclose": (herr_t, [hid_t]),
        "H5Screate_simple": (hid_t, [ctypes.c_int, ctypes.POINTER(hsize_t), ctypes.POINTER(hsize_t)]),
        "H5Sget_simple_extent_ndims": (ctypes.c_int, [hid_t]),
        "H5Sget_simple_extent_dims": (ctypes.c_int, [hid_t, ctypes.POINTER(hsize_t), ctypes.POINTER(hsize_t)]),
        "H5Sclose": (herr_t, [hid_t]),
        "H5Dcreate2": (hid_t, [hid_t, ctypes.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
        "H5Dopen2": (hid_t, [hid_t, ctypes.c_char_p, hid_t]),
        "H5Dwrite": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, ctypes.c_void_p]),
        "H5Dread": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, ctypes.c_void_p]),
        "H5Dget_space": (hid_t, [hid_t]),
        "H5Dget_type": (hid_t, [hid_t]),
        "H5Dclose": (herr_t, [hid_t]),
        "H5Tget_class": (ctypes.c_int, [hid_t]),
        "H5Tget_size": (ctypes.c_size_t, [hid_t]),
        "H5Tget_sign": (ctypes.c_int, [hid_t]),
        "H5Tclose": (herr_t, [hid_t]),
        "H5Lexists": (ctypes.c_int, [hid_t, ctypes.c_char_p, hid_t]),
        "H5Ldelete": (herr_t, [hid_t, ctypes.c_char_p, hid_t]),
    }
    for name, (res, args) in sig.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise H5Unavailable(f"{lib._name} does not export {name}: not an HDF5 >= 1.10 C library")
        fn.restype, fn.argtypes = res, args
    if lib.H5open() < 0:
        raise H5Unavailable(f"H5open failed in {lib._name}")
    lib.H5Eset_auto2(0, None, None)          # errors come back as negative return values -> H5Error here, not a stack dump on stderr
    _lib = lib
    return lib


def _link_visitor(lib):
    """The library's link visitor: H5Lvisit (an exported symbol up to 1.10), H5Lvisit2 or H5Lvisit1 (1.12+, where H5Lvisit is a macro).
    The three differ in the info struct handed to the callback, which the callback here ignores. Resolved on first use (keys())."""
    for name in ("H5Lvisit", "H5Lvisit2", "H5Lvisit1"):
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype, fn.argtypes = herr_t, [hid_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
            return fn
    raise H5Unavailable(f"{lib._name} exports none of H5Lvisit / H5Lvisit2 / H5Lvisit1: the keys of a store cannot be listed")


def available():
    try:
        load()
        return True
    except H5Unavailable:
        return False


def library_version():
    a, b, c = ctypes.c_uint(), ctypes.c_uint(), ctypes.c_uint()
    load().H5get_libversion(ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
    return (a.value, b.value, c.value)


def _gid(name):
    """Value of one of the library's global identifiers (H5T_NATIVE_*_g, H5P_CLS_*_ID_g; valid after H5open)."""
    return hid_t.in_dll(load(), name).value


_NATIVE = {
    np.dtype(np.float32): "H5T_NATIVE_FLOAT_g", np.dtype(np.float64): "H5T_NATIVE_DOUBLE_g",
    np.dtype(np.int8): "H5T_NATIVE_INT8_g", np.dtype(np.uint8): "H5T_NATIVE_UINT8_g",
    np.dtype(np.int16): "H5T_NATIVE_INT16_g", np.dtype(np.uint16): "H5T_NATIVE_UINT16_g",
    np.dtype(np.int32): "H5T_NATIVE_INT32_g", np.dtype(np.uint32): "H5T_NATIVE_UINT32_g",
    np.dtype(np.int64): "H5T_NATIVE_INT64_g", np.dtype(np.uint64): "H5T_NATIVE_UINT64_g",
}


def _check(rc, what):
    if rc < 0:
        raise H5Error(f"HDF5: {what} failed")
    return rc


class H5Store:
    """``with H5Store(path, "w") as hf: hf[key] = array`` / ``H5Store(path)[key]`` / ``.keys()`` - the part of h5py.File the
    reference's result loop uses (interfaceome/apply_model.py:53, :76). Mode "w" truncates, "r" (default) opens read-only."""

    def __init__(self, path, mode="r"):
        lib = load()
        self._lib = lib
        self._id = -1
        self.path = os.fspath(path)
        self.mode = mode
        if mode == "w":
            self._id = lib.H5Fcreate(self.path.encode(), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)
        elif mode == "r":
            self._id = lib.H5Fopen(self.path.encode(), H5F_ACC_RDONLY, H5P_DEFAULT)
        else:
            raise ValueError("mode must be 'r' or 'w'")
        if self._id < 0:
            raise H5Error(f"HDF5: cannot {'create' if mode == 'w' else 'open'} {self.path}")
        self._lcpl = -1
        if mode == "w":
            self._lcpl = _check(lib.H5Pcreate(_gid("H5P_CLS_LINK_CREATE_ID_g")), "H5Pcreate(link creation)")
            _check(lib.H5Pset_create_intermediate_group(self._lcpl, 1), "H5Pset_create_intermediate_group")

    # ---- writing
    def __setitem__(self, key, value):
        if self.mode != "w":
            raise H5Error("store is read-only")
        a = np.ascontiguousarray(value)
        if a.dtype == np.bool_:
            a = a.astype(np.uint8)
        if a.dtype not in _NATIVE:
            raise TypeError(f"dtype {a.dtype} has no HDF5 mapping here (float32/64 and the 8..64-bit integers do)")
        lib, name = self._lib, self._name(key)
        if lib.H5Lexists(self._id, name, H5P_DEFAULT) > 0:
            raise H5Error(f"HDF5: name already exists: {key}")          # h5py: "Unable to create dataset (name already exists)"
        dims = (hsize_t * max(a.ndim, 1))(*a.shape)
        space = _check(lib.H5Screate_simple(a.ndim, dims if a.ndim else None, None), "H5Screate_simple")
        try:
            tid = _gid(_NATIVE[a.dtype])
            ds = _check(lib.H5Dcreate2(self._id, name, tid, space, self._lcpl, H5P_DEFAULT, H5P_DEFAULT), f"H5Dcreate2({key})")
            try:
                if a.size:
                    _check(lib.H5Dwrite(ds, tid, H5S_ALL, H5S_ALL, H5P_DEFAULT, a.ctypes.data_as(ctypes.c_void_p)), f"H5Dwrite({key})")
            finally:
                lib.H5Dclose(ds)
        finally:
            lib.H5Sclose(space)

    # ---- reading
    def __getitem__(self, key):
        lib = self._lib
        ds = lib.H5Dopen2(self._id, self._name(key), H5P_DEFAULT)
        if ds < 0:
            raise KeyError(key)
        try:
            space = _check(lib.H5Dget_space(ds), "H5Dget_space")
            try:
                nd = _check(lib.H5Sget_simple_extent_ndims(space), "H5Sget_simple_extent_ndims")
                dims = (hsize_t * max(nd, 1))()
                if nd:
                    _check(lib.H5Sget_simple_extent_dims(space, dims, None), "H5Sget_simple_extent_dims")
                shape = tuple(int(dims[i]) for i in range(nd))
            finally:
                lib.H5Sclose(space)
            ft = _check(lib.H5Dget_type(ds), "H5Dget_type")
            try:
                cls, size = lib.H5Tget_class(ft), int(lib.H5Tget_size(ft))
                if cls == H5T_FLOAT and size in (4, 8):
                    dt = np.dtype(np.float32 if size == 4 else np.float64)
                elif cls == H5T_FLOAT and size == 2:
                    dt = np.dtype(np.float32)                  # half-precision files are widened by the library
                elif cls == H5T_INTEGER and size in (1, 2, 4, 8):
                    dt = np.dtype(("u" if lib.H5Tget_sign(ft) == H5T_SGN_NONE else "i") + str(size))
                else:
                    raise H5Error(f"HDF5: dataset {key}: type class {cls} of {size} bytes is not read here")
            finally:
                lib.H5Tclose(ft)
            out = np.empty(shape, dt)
            if out.size:          # memory type = the native type of `dt`: the library converts byte order / width
                _check(lib.H5Dread(ds, _gid(_NATIVE[dt]), H5S_ALL, H5S_ALL, H5P_DEFAULT, out.ctypes.data_as(ctypes.c_void_p)), f"H5Dread({key})")
            return out
        finally:
            lib.H5Dclose(ds)

    def __contains__(self, key):
        lib = self._lib
        # H5Lexists wants every intermediate link to exist: walk the path
        parts = [p for p in str(key).split("/") if p]
        for i in range(1, len(parts) + 1):
            if lib.H5Lexists(self._id, "/".join(parts[:i]).encode(), H5P_DEFAULT) <= 0:
                return False
        return bool(parts)

    def keys(self):
        """Full names of all DATASETS in the file, in name order (h5py's File.visit order), without the leading '/'."""
        lib, names = self._lib, []
        cb_t = ctypes.CFUNCTYPE(herr_t, hid_t, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p)

        def visit(_group, name, _info, _data):
            names.append(name.decode())
            return 0

        cb = cb_t(visit)
        _check(_link_visitor(lib)(self._id, H5_INDEX_NAME, H5_ITER_NATIVE, ctypes.cast(cb, ctypes.c_void_p), None), "H5Lvisit")
        out = []
        for n in names:                        # links to groups are visited too: keep what opens as a dataset
            ds = lib.H5Dopen2(self._id, n.encode(), H5P_DEFAULT)
            if ds >= 0:
                lib.H5Dclose(ds)
                out.append(n)
        return sorted(out)

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self.keys())

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    @staticmethod
    def _name(key):
        name = str(key)
        if not name.strip("/"):
            raise KeyError(key)
        return name.encode()

    # ---- life cycle
    def flush(self):
        if self._id >= 0:
            _check(self._lib.H5Fflush(self._id, 1), "H5Fflush")          # scope 1 = H5F_SCOPE_GLOBAL

    def close(self):
        if self._id >= 0:
            if self._lcpl >= 0:
                self._lib.H5Pclose(self._lcpl)
                self._lcpl = -1
            rc = self._lib.H5Fclose(self._id)
            self._id = -1
            _check(rc, f"H5Fclose({self.path})")

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
