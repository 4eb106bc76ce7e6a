"""pesto_amd: MI355X-native (gfx950) drop-in for PeSTo's geometric-transformer forward pass.

Only the hot path ``Model(config).forward(X, ids_topk, q, M)`` (reference model/model.py:32-52) lives
here: a Python host mirror of the reference Module API over a C-ABI HIP library. See DESIGN.md.
"""
from .config import CONFIGS, config_i_v3_0, config_i_v3_1, config_i_v4_0, config_i_v4_1, config_model  # noqa: F401

__all__ = ["Model", "CONFIGS", "config_model", "config_i_v4_1", "config_i_v4_0", "config_i_v3_0", "config_i_v3_1"]


def __getattr__(name):  # lazy: importing the package must not need torch or the built library
    if name == "Model":
        from .model import Model
        return Model
    raise AttributeError(name)
