"""ctypes binding of libpesto_hip.so (the C ABI declared in include/pesto_hip.h).

The library is built in-tree by ``python -m pesto_amd.csrc.build`` (hipcc --offload-arch=gfx950) and
must be present: there is NO fallback path - a missing library raises at first use.
"""
import ctypes
import os

from .config import MAX_LAYERS, normalise

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PESTO_LIB") or os.path.join(_HERE, "csrc", "libpesto_hip.so")   # PESTO_LIB: developer builds

PTR_HOST, PTR_DEVICE = 0, 1
IDS_INT32, IDS_INT64, IDS_UINT16 = 32, 64, 16
IDS_NARROW = 0x100      # OR-ed to IDS_INT32 / IDS_INT64 (pesto_forward_batch_submit): staged as uint16, narrowed and range-checked by the packer
BATCH_COLLATED, BATCH_INDEPENDENT = 0, 1
# enum pesto_precision
PRECISIONS = {"auto": 0, "f16_split": 1, "fp32": 2}
ERR_RANGE = -5


class PestoConfig(ctypes.Structure):
    """struct pesto_config (include/pesto_hip.h)."""
    _fields_ = [
        ("n0", ctypes.c_int32),
        ("n_layers", ctypes.c_int32),
        ("nn", ctypes.c_int32 * MAX_LAYERS),
        ("n_out", ctypes.c_int32),
        ("em_depth", ctypes.c_int32),
        ("dm_depth", ctypes.c_int32),
        ("precision", ctypes.c_int32),
    ]


def precision_code(precision):
    try:
        return PRECISIONS[str(precision).lower()]
    except KeyError:
        raise ValueError(f"precision must be one of {sorted(PRECISIONS)}, got {precision!r}") from None


def make_c_config(config, precision="auto"):
    c = normalise(config)
    cc = PestoConfig()
    cc.n0 = c["em"]["N0"]
    cc.n_layers = len(c["sum"])
    for i, l in enumerate(c["sum"]):
        cc.nn[i] = l["nn"]
    cc.n_out = c["dm"]["N2"]
    cc.em_depth = c["em_depth"]
    cc.dm_depth = c["dm_depth"]
    cc.precision = precision_code(precision)
    return cc


# every symbol include/pesto_hip.h declares (tests/test_abi.py checks the built library exports them all)
ABI_SYMBOLS = [
    "pesto_last_error", "pesto_blob_size", "pesto_create", "pesto_destroy", "pesto_forward",
    "pesto_workspace_bytes", "pesto_synchronize", "pesto_set_timing", "pesto_get_timing",
    "pesto_stage_embed", "pesto_stage_unpack", "pesto_stage_layer", "pesto_stage_pool", "pesto_knn_collate",
    "pesto_forward_frames", "pesto_postprocess", "pesto_forward_batch", "pesto_get_kernel_timing",
    "pesto_set_precision", "pesto_get_status", "pesto_debug_select", "pesto_forward_structures",
    "pesto_mask_to_segments", "pesto_debug_edge_mode", "pesto_forward_batch_submit", "pesto_forward_batch_wait",
    "pesto_set_async_auto", "pesto_debug_host_only", "pesto_knn_tie_rows", "pesto_set_auto_state_limit", "pesto_set_auto_pad_trigger",
    "pesto_get_auto_counters",
]

_lib = None


class PestoError(RuntimeError):
    """code: the negative pesto_status the library returned (None when raised by the Python layer)."""
    code = None


def load():
    """Load libpesto_hip.so (once). Raises PestoError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PestoError(f"{LIB_PATH} not found: build it with `python -m pesto_amd.csrc.build` "
                         "(there is no CPU/PyTorch fallback for the forward pass)")
    # ONE HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 / libhsa-runtime64. If this
    # library pulled in /opt/rocm's copy first, torch would later load a second runtime and one of the two
    # would see no device. Importing torch first makes the dynamic loader resolve our DT_NEEDED
    # libamdhip64.so.7 to the copy torch already mapped (SONAME match).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    c_p, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
    P = ctypes.POINTER
    lib.pesto_last_error.restype = ctypes.c_char_p
    lib.pesto_last_error.argtypes = []
    lib.pesto_blob_size.argtypes = [P(PestoConfig), P(i64)]
    lib.pesto_create.argtypes = [P(PestoConfig), c_p, i64, ctypes.c_int, P(c_p)]
    lib.pesto_destroy.argtypes = [c_p]
    lib.pesto_forward.argtypes = [c_p, i64, i64, i32, c_p, c_p, i32, c_p, c_p, c_p, i32, c_p]
    lib.pesto_forward_structures.argtypes = [c_p, i64, i64, i32, i32, c_p, c_p, c_p, i32, c_p, c_p, c_p, i32, c_p]
    lib.pesto_forward_frames.argtypes = [c_p, i64, i64, i32, i64, c_p, i64, i64, c_p, i32, c_p, c_p, c_p, i32, i32, c_p]
    lib.pesto_forward_batch.argtypes = [c_p, i32, c_p, c_p, c_p, c_p, c_p, i32, c_p, c_p, c_p, i32, c_p]
    lib.pesto_forward_batch_submit.argtypes = [c_p, i32, c_p, c_p, c_p, c_p, c_p, i32, c_p, c_p, i32, c_p, c_p, c_p, i32, P(i32)]
    lib.pesto_forward_batch_wait.argtypes = [c_p, i32]
    lib.pesto_set_precision.argtypes = [c_p, i32]
    lib.pesto_set_async_auto.argtypes = [c_p, i32]
    lib.pesto_set_auto_state_limit.argtypes = [c_p, ctypes.c_float]
    lib.pesto_set_auto_pad_trigger.argtypes = [c_p, i32]
    lib.pesto_debug_host_only.argtypes = [c_p, i32]
    lib.pesto_get_status.argtypes = [c_p, P(i32), P(i64), P(i64)]
    lib.pesto_get_auto_counters.argtypes = [c_p, P(i64), P(i64)]
    lib.pesto_debug_select.argtypes = [c_p, i32, i32]
    lib.pesto_debug_edge_mode.argtypes = [c_p, i32]
    lib.pesto_mask_to_segments.argtypes = [c_p, i64, i64, c_p, c_p, i32, c_p]
    lib.pesto_postprocess.argtypes = [c_p, i64, i64, c_p, c_p, c_p, c_p, i32, c_p]
    lib.pesto_workspace_bytes.argtypes = [c_p, i64, i64, P(i64)]
    lib.pesto_synchronize.argtypes = [c_p]
    lib.pesto_set_timing.argtypes = [c_p, i32]
    lib.pesto_get_timing.argtypes = [c_p, P(ctypes.c_double), P(ctypes.c_double), P(i32)]
    lib.pesto_get_kernel_timing.argtypes = [c_p, P(ctypes.c_double), P(i32)]
    lib.pesto_knn_collate.argtypes = [c_p, i64, i32, c_p, c_p, i32, c_p, i32, i32, c_p]
    lib.pesto_knn_tie_rows.argtypes = [c_p, i64, i32, c_p, c_p, i32, c_p, i32, c_p, i32, c_p]
    lib.pesto_stage_embed.argtypes = [c_p, i64, c_p, c_p]
    lib.pesto_stage_unpack.argtypes = [c_p, i64, i32, c_p, c_p, i32, c_p, c_p]
    lib.pesto_stage_layer.argtypes = [c_p, i32, c_p, c_p]
    lib.pesto_stage_pool.argtypes = [c_p, i64, i64, c_p, c_p, c_p, c_p, c_p, c_p]
    for name in ABI_SYMBOLS:
        if name != "pesto_last_error":
            getattr(lib, name).restype = ctypes.c_int
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().pesto_last_error()
        err = PestoError(f"libpesto_hip error {rc}: {msg.decode() if msg else '?'}")
        err.code = rc
        raise err
