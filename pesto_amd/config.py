"""Model configuration dicts, in the schema the reference's ``Model(config)`` consumes.

Schema (reference: model/config.py:25-63 and the per-run copies under model/save/<run>/config.py):
  {"em": {N0, N1}, "sum": [{Ns, Nh, Nk, nn}, ...], "spl": {N0, N1, Nh}, "dm": {N0, N1, N2}}
plus two keys the reference encodes in the per-run model.py instead of the dict
(model/save/i_v3_1_2021-05-28_12-40/model.py:10-22): ``em_depth`` / ``dm_depth`` = number of
Linear layers in the embedding / decoding MLP (3 everywhere except i_v3_1, which uses 1).
A plain reference dict (without those keys) is accepted everywhere and means depth 3.
"""

MAX_LAYERS = 64  # must match PESTO_MAX_LAYERS in include/pesto_hip.h


def _sum(pattern, Ns=32, Nh=2, Nk=3):
    layers = []
    for nn, count in pattern:
        layers += [{"Ns": Ns, "Nh": Nh, "Nk": Nk, "nn": nn} for _ in range(count)]
    return layers


def make_config(n0, pattern, n_out=5, em_depth=3, dm_depth=3):
    return {
        "em": {"N0": n0, "N1": 32},
        "sum": _sum(pattern),
        "spl": {"N0": 32, "N1": 32, "Nh": 4},
        "dm": {"N0": 32, "N1": 32, "N2": n_out},
        "em_depth": em_depth,
        "dm_depth": dm_depth,
    }


# i_v4_1: the flagship, 32 layers, element one-hot (30)          model/config.py:25-63
config_i_v4_1 = make_config(30, [(8, 8), (16, 8), (32, 8), (64, 8)])
# i_v4_0: 16 layers, element one-hot                              model/save/i_v4_0_*/config.py
config_i_v4_0 = make_config(30, [(8, 4), (16, 4), (32, 4), (64, 4)])
# i_v3_0: 16 layers, element+resname+atom-name one-hots (123)     model/save/i_v3_0_*/config.py:26-47
config_i_v3_0 = make_config(123, [(8, 4), (16, 4), (32, 4), (64, 4)])
# i_v3_1: i_v3_0 with single-Linear em and dm, one output         model/save/i_v3_1_*/{config,model}.py
config_i_v3_1 = make_config(123, [(8, 4), (16, 4), (32, 4), (64, 4)], n_out=1, em_depth=1, dm_depth=1)

CONFIGS = {"i_v4_1": config_i_v4_1, "i_v4_0": config_i_v4_0, "i_v3_0": config_i_v3_0, "i_v3_1": config_i_v3_1}

# name used by the reference's notebooks: ``from config import config_model``
config_model = config_i_v4_1


def normalise(config):
    """Validate a reference-style dict and fill the two depth keys. Returns a new dict."""
    c = {k: (dict(v) if isinstance(v, dict) else v) for k, v in config.items()}
    c["sum"] = [dict(l) for l in config["sum"]]
    c.setdefault("em_depth", 3)
    c.setdefault("dm_depth", 3)
    if len(c["sum"]) == 0 or len(c["sum"]) > MAX_LAYERS:
        raise ValueError(f"number of layers must be in 1..{MAX_LAYERS}")
    for l in c["sum"]:
        if (l["Ns"], l["Nh"], l["Nk"]) != (32, 2, 3):
            raise ValueError("the HIP path is specialised for Ns=32, Nh=2, Nk=3 (all reference runs)")
        if l["nn"] not in (8, 16, 32, 64):
            raise ValueError("nn must be one of 8, 16, 32, 64")
    if c["em"]["N1"] != 32 or c["spl"] != {"N0": 32, "N1": 32, "Nh": 4} or c["dm"]["N0"] != 32 or c["dm"]["N1"] != 32:
        raise ValueError("the HIP path is specialised for 32-wide em/spl/dm (all reference runs)")
    if c["em_depth"] not in (1, 3) or c["dm_depth"] not in (1, 3):
        raise ValueError("em_depth/dm_depth must be 1 or 3")
    if not (1 <= c["dm"]["N2"] <= 32) or not (1 <= c["em"]["N0"] <= 512):
        raise ValueError("dm.N2 must be in 1..32 and em.N0 in 1..512")
    return c
