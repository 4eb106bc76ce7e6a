"""C-ABI surface checks that need no GPU: the library loads, exports every symbol include/pesto_hip.h declares,
agrees with the Python blob schema, and fails LOUDLY (no fallback) when no device is present."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, weights
from pesto_amd import _lib
from pesto_amd.config import CONFIGS
from pesto_amd.weights import blob_size, flatten_state_dict


def test_header_symbols_exported():
    hdr = open(os.path.join(ROOT, "include", "pesto_hip.h")).read()
    declared = set(re.findall(r"\b(pesto_[a-z_]+)\s*\(", hdr))
    assert declared == set(_lib.ABI_SYMBOLS)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name


def test_config_struct_layout():
    # struct pesto_config: n0, n_layers, nn[64], n_out, em_depth, dm_depth (all int32)
    assert ctypes.sizeof(_lib.PestoConfig) == 4 * (2 + 64 + 3)


@pytest.mark.parametrize("tag", sorted(CONFIGS))
def test_blob_size_agrees(tag):
    lib = _lib.load()
    n = ctypes.c_int64()
    cc = _lib.make_c_config(CONFIGS[tag])
    assert lib.pesto_blob_size(ctypes.byref(cc), ctypes.byref(n)) == 0
    assert n.value == blob_size(CONFIGS[tag])


def test_invalid_config_rejected():
    lib = _lib.load()
    cc = _lib.make_c_config(CONFIGS["i_v4_0"])
    cc.nn[3] = 12
    n = ctypes.c_int64()
    assert lib.pesto_blob_size(ctypes.byref(cc), ctypes.byref(n)) == -1
    assert b"pesto_config" in lib.pesto_last_error()


def test_create_rejects_wrong_blob_size():
    lib = _lib.load()
    cc = _lib.make_c_config(CONFIGS["i_v4_0"])
    blob = np.zeros(10, np.float32)
    h = ctypes.c_void_p()
    rc = lib.pesto_create(ctypes.byref(cc), blob.ctypes.data, blob.size, 0, ctypes.byref(h))
    assert rc == -1 and not h.value


def test_no_gpu_fails_loudly(gpu_available):
    """Without a device the product path must raise, never fall back to a CPU implementation."""
    if gpu_available:
        pytest.skip("a GPU is present")
    from pesto_amd import Model
    m = Model(CONFIGS["i_v4_0"])
    m.load_state_dict(weights("i_v4_0"))
    X = np.zeros((70, 3), np.float32)
    with pytest.raises(_lib.PestoError):
        m.forward_segments(X, np.zeros((70, 64), np.int64), np.zeros((70, 30), np.float32), np.zeros(70, np.int32), 1)


def test_flatten_is_strict():
    sd = dict(weights("i_v4_0"))
    flatten_state_dict(CONFIGS["i_v4_0"], sd)
    bad = dict(sd); bad.pop("sum.3.su.evm.2.bias")
    with pytest.raises(KeyError):
        flatten_state_dict(CONFIGS["i_v4_0"], bad)
    bad = dict(sd); bad["dm.4.weight"] = np.zeros((4, 32), np.float32)
    with pytest.raises(ValueError):
        flatten_state_dict(CONFIGS["i_v4_0"], bad)
    with pytest.raises(KeyError):
        flatten_state_dict(CONFIGS["i_v4_1"], sd)   # 32-layer config, 16-layer weights
