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
    # struct pesto_config: n0, n_layers, nn[64], n_out, em_depth, dm_depth, precision (all int32); the header's field order
    hdr = open(os.path.join(ROOT, "include", "pesto_hip.h")).read()
    body = re.search(r"typedef struct pesto_config \{(.*?)\} pesto_config;", hdr, re.S).group(1)
    fields = re.findall(r"int32_t\s+(\w+)", body)
    assert fields == [f[0] for f in _lib.PestoConfig._fields_]
    assert ctypes.sizeof(_lib.PestoConfig) == 4 * (2 + 64 + 4)
    # enum pesto_precision as the Python layer spells it
    for name, code in (("AUTO", "auto"), ("F16_SPLIT", "f16_split"), ("FP32", "fp32")):
        assert int(re.search(rf"PESTO_PRECISION_{name} = (\d+)", hdr).group(1)) == _lib.PRECISIONS[code]
    assert int(re.search(r"PESTO_ERR_RANGE = (-\d+)", hdr).group(1)) == _lib.ERR_RANGE


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


def test_invalid_precision_rejected():
    lib = _lib.load()
    cc = _lib.make_c_config(CONFIGS["i_v4_0"], "fp32")
    assert cc.precision == 2
    cc.precision = 7
    n = ctypes.c_int64()
    assert lib.pesto_blob_size(ctypes.byref(cc), ctypes.byref(n)) == -1
    with pytest.raises(ValueError):
        _lib.make_c_config(CONFIGS["i_v4_0"], "bf16")
    from pesto_amd import Model
    with pytest.raises(ValueError):
        Model(CONFIGS["i_v4_0"], precision="fastest")
    assert Model(CONFIGS["i_v4_0"]).precision == "auto"                       # the default policy
    assert lib.pesto_set_precision(None, 0) == -1 and lib.pesto_debug_select(None, 0, 0) == -1


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
    # strict=False (torch's meaning): unexpected keys are ignored, missing ones still raise
    extra = dict(sd); extra["optimizer.step"] = np.zeros(1, np.float32)
    with pytest.raises(KeyError):
        flatten_state_dict(CONFIGS["i_v4_0"], extra)
    assert np.array_equal(flatten_state_dict(CONFIGS["i_v4_0"], extra, strict=False), flatten_state_dict(CONFIGS["i_v4_0"], sd))
    from pesto_amd import Model
    m = Model(CONFIGS["i_v4_0"])
    with pytest.raises(KeyError):
        m.load_state_dict(extra)
    m.load_state_dict(extra, strict=False)
    bad = dict(extra); bad.pop("sum.3.su.evm.2.bias")
    with pytest.raises(KeyError):
        m.load_state_dict(bad, strict=False)
