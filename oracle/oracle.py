"""ctypes wrapper of the CPU oracle (oracle/pesto_oracle.c).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (pesto_amd/) never imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

from pesto_amd._lib import PestoConfig, make_c_config
from pesto_amd.weights import flatten_state_dict

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libpesto_oracle.so")
_SO_WIDE = os.path.join(_HERE, "libpesto_oracle_wide.so")      # same source, double accumulators (header of pesto_oracle.c)
_libs = {}


def build(force=False):
    src = os.path.join(_HERE, "pesto_oracle.c")
    if force or any(not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src) for so in (_SO, _SO_WIDE)):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"] if force else ["make", "-C", _HERE, "-s"])
    return _SO


def _load(wide=False):
    if wide not in _libs:
        build()
        lib = ctypes.CDLL(_SO_WIDE if wide else _SO)
        c_p, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
        P = ctypes.POINTER
        lib.oracle_blob_size.argtypes = [P(PestoConfig), P(i64)]
        lib.oracle_create.argtypes = [P(PestoConfig), c_p, i64, P(c_p)]
        lib.oracle_destroy.argtypes = [c_p]
        lib.oracle_destroy.restype = None
        lib.oracle_set_threads.argtypes = [ctypes.c_int]
        lib.oracle_set_threads.restype = None
        lib.oracle_embed.argtypes = [c_p, i64, c_p, c_p]
        lib.oracle_unpack.argtypes = [i64, ctypes.c_int, c_p, c_p, c_p, c_p, c_p]
        lib.oracle_layer.argtypes = [c_p, ctypes.c_int, i64, ctypes.c_int, c_p, c_p, c_p, c_p, c_p]
        lib.oracle_pool.argtypes = [c_p, i64, i64, c_p, c_p, c_p, c_p, c_p, c_p]
        lib.oracle_forward.argtypes = [c_p, i64, i64, ctypes.c_int, c_p, c_p, c_p, c_p, c_p, c_p, c_p, ctypes.c_int]
        _libs[wide] = lib
    return _libs[wide]


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def set_threads(n):
    """OpenMP team size of the following oracle calls (n < 1: all cores)."""
    for lib in list(_libs.values()) or [_load()]:
        lib.oracle_set_threads(int(n))


def blob_size(config):
    n = ctypes.c_int64()
    cc = make_c_config(config)
    assert _load().oracle_blob_size(ctypes.byref(cc), ctypes.byref(n)) == 0
    return n.value


class OracleModel:
    """CPU oracle with the reference Module's call shape: OracleModel(config, state_dict)(X, ids_topk, q0, M).
    wide=True: the build with double accumulators (float32 storage) - the checker for ill-conditioned inputs."""

    def __init__(self, config, state_dict, wide=False):
        self.config = config
        self.cc = make_c_config(config)
        self._l = _load(bool(wide))
        blob = _f32(flatten_state_dict(config, state_dict))
        h = ctypes.c_void_p()
        rc = self._l.oracle_create(ctypes.byref(self.cc), _ptr(blob), blob.size, ctypes.byref(h))
        if rc != 0:
            raise RuntimeError(f"oracle_create failed: {rc}")
        self.h = h
        self.n_out = self.cc.n_out

    def __del__(self):
        if getattr(self, "h", None) and getattr(self, "_l", None) is not None:
            self._l.oracle_destroy(self.h)
            self.h = None

    # ---- stages
    def embed(self, q0):
        q0 = _f32(q0)
        out = np.empty((q0.shape[0], 32), np.float32)
        self._l.oracle_embed(self.h, q0.shape[0], _ptr(q0), _ptr(out))
        return out

    @staticmethod
    def unpack(X, ids_topk):
        X, ids = _f32(X), _i32(ids_topk)
        n, k = ids.shape
        ids_s = np.empty((n + 1, k), np.int32)
        D = np.empty((n + 1, k), np.float32)
        R = np.empty((n + 1, k, 3), np.float32)
        _load().oracle_unpack(n, k, _ptr(X), _ptr(ids), _ptr(ids_s), _ptr(D), _ptr(R))
        return ids_s, D, R

    def layer(self, layer, ids_s, D, R, q, p):
        q, p = _f32(q).copy(), _f32(p).copy()
        ids_s, D, R = _i32(ids_s), _f32(D), _f32(R)
        rc = self._l.oracle_layer(self.h, layer, q.shape[0], ids_s.shape[1], _ptr(ids_s), _ptr(D), _ptr(R), _ptr(q), _ptr(p))
        assert rc == 0
        return q, p

    def pool(self, q, p, res_of_atom, R):
        q, p, roa = _f32(q), _f32(p), _i32(res_of_atom)
        qr = np.empty((R, 32), np.float32)
        pr = np.empty((R, 3, 32), np.float32)
        z = np.empty((R, self.n_out), np.float32)
        rc = self._l.oracle_pool(self.h, q.shape[0], R, _ptr(q), _ptr(p), _ptr(roa), _ptr(qr), _ptr(pr), _ptr(z))
        if rc != 0:
            raise RuntimeError("oracle_pool: empty residue")
        return qr, pr, z

    # ---- whole forward
    def forward_segments(self, X, ids_topk, q0, res_of_atom, R, stop_after=-1, return_state=False):
        X, ids, q0, roa = _f32(X), _i32(ids_topk), _f32(q0), _i32(res_of_atom)
        n, k = ids.shape
        z = np.empty((R, self.n_out), np.float32)
        qs = np.empty((n + 1, 32), np.float32) if return_state else None
        ps = np.empty((n + 1, 3, 32), np.float32) if return_state else None
        rc = self._l.oracle_forward(self.h, n, R, k, _ptr(X), _ptr(ids), _ptr(q0), _ptr(roa), _ptr(z),
                                    _ptr(qs), _ptr(ps), stop_after)
        if rc != 0:
            raise RuntimeError(f"oracle_forward failed: {rc}")
        return (z, qs, ps) if return_state else z

    def __call__(self, X, ids_topk, q0, M):
        from pesto_amd.topology import mask_to_segments
        roa, R = mask_to_segments(M)
        return self.forward_segments(np.asarray(X), np.asarray(ids_topk), np.asarray(q0), roa, R)
