"""Weight-blob schema shared by the HIP library, the C oracle and the Python host layer.

The reference keeps weights in a torch ``state_dict`` (model/model.py:10-30,
src/model_operations.py:27-85, :172-195, :218-223).  The C ABI takes ONE flat fp32 blob holding
every floating-point tensor of that dict in ``state_dict`` iteration order, skipping the two
non-learned entries per layer (``sum.<l>.m_nn`` int64 arange, ``sum.<l>.su.sdk`` = float32(sqrt(Nk))).
``weight`` tensors stay ``[out, in]`` row-major exactly as torch stores them.
The same table is restated in C in pesto_amd/csrc/pesto_schema.h (one source of truth per
language; tests/test_weights.py checks they agree through pesto_blob_size()).
"""
import numpy as np

from .config import normalise


def _mlp(prefix, dims, bias_last=True):
    """Keys of a Sequential(Linear, ELU, Linear, ELU, Linear): indices 0, 2, 4."""
    out = []
    for i, (n_in, n_out) in enumerate(zip(dims[:-1], dims[1:])):
        out.append((f"{prefix}.{2 * i}.weight", (n_out, n_in)))
        out.append((f"{prefix}.{2 * i}.bias", (n_out,)))
    return out


def blob_schema(config):
    """List of (state_dict key, shape) in blob order."""
    c = normalise(config)
    n0, s = c["em"]["N0"], 32
    n2 = c["dm"]["N2"]
    keys = []
    keys += _mlp("em", [n0, s, s, s] if c["em_depth"] == 3 else [n0, s])
    for li, l in enumerate(c["sum"]):
        pre = f"sum.{li}.su"
        nh, nk = l["Nh"], l["Nk"]
        keys += _mlp(pre + ".nqm", [2 * s, s, s, 2 * nk * nh])
        keys += _mlp(pre + ".eqkm", [6 * s + 1, s, s, nk])
        keys += _mlp(pre + ".epkm", [6 * s + 1, s, s, 3 * nk])
        keys += _mlp(pre + ".evm", [6 * s + 1, 2 * s, 2 * s, 2 * s])
        keys += _mlp(pre + ".qpm", [nh * s, s, s, s])
        keys.append((pre + ".ppm.0.weight", (s, nh * s)))
    ph = c["spl"]["Nh"]
    keys += _mlp("spl.sam", [2 * s, s, s, 2 * ph])
    keys += _mlp("spl.zdm", [ph * s, s, s, s])
    keys.append(("spl.zdm_vec.0.weight", (s, ph * s)))
    keys += _mlp("dm", [2 * s, s, s, n2] if c["dm_depth"] == 3 else [2 * s, n2])
    return keys


def blob_size(config):
    return int(sum(int(np.prod(shape)) for _, shape in blob_schema(config)))


def _to_numpy(v):
    if hasattr(v, "detach"):
        v = v.detach().cpu().numpy()
    return np.asarray(v)


def flatten_state_dict(config, state_dict, strict=True):
    """state_dict (torch tensors or numpy arrays, reference key names) -> flat float32 blob.

    Mirrors ``Module.load_state_dict`` strictness: a missing key or a shape mismatch raises
    KeyError / ValueError; ``m_nn`` and ``sdk`` entries are checked if present, then dropped.
    ``strict=False`` tolerates unexpected keys (a missing parameter still raises: the blob must be complete).
    """
    c = normalise(config)
    parts = []
    for key, shape in blob_schema(c):
        if key not in state_dict:
            raise KeyError(f"missing key in state_dict: {key}")
        a = _to_numpy(state_dict[key])
        if tuple(a.shape) != tuple(shape):
            raise ValueError(f"size mismatch for {key}: got {tuple(a.shape)}, expected {tuple(shape)}")
        parts.append(np.ascontiguousarray(a, dtype=np.float32).ravel())
    for li, l in enumerate(c["sum"]):
        k = f"sum.{li}.m_nn"
        if k in state_dict:
            m = _to_numpy(state_dict[k])
            if m.shape != (l["nn"],) or not np.array_equal(m, np.arange(l["nn"])):
                raise ValueError(f"{k} must be arange({l['nn']}) (model_operations.py:223)")
        k = f"sum.{li}.su.sdk"
        if k in state_dict:
            sdk = float(_to_numpy(state_dict[k]))
            if abs(sdk - float(np.sqrt(np.float32(l["Nk"])))) > 1e-6:
                raise ValueError(f"{k} must be sqrt(Nk) (model_operations.py:85)")
    known = {k for k, _ in blob_schema(c)}
    extra = [k for k in state_dict if k not in known and not (k.endswith(".m_nn") or k.endswith(".su.sdk"))]
    if extra and strict:
        raise KeyError(f"unexpected key(s) in state_dict: {extra[:4]}")
    return np.concatenate(parts)


def unflatten_blob(config, blob):
    """Inverse of flatten_state_dict (numpy views), adding m_nn / sdk so the dict is loadable by the reference."""
    c = normalise(config)
    out, off = {}, 0
    for key, shape in blob_schema(c):
        n = int(np.prod(shape))
        out[key] = blob[off:off + n].reshape(shape)
        off += n
    for li, l in enumerate(c["sum"]):
        out[f"sum.{li}.m_nn"] = np.arange(l["nn"], dtype=np.int64)
        out[f"sum.{li}.su.sdk"] = np.sqrt(np.float32(l["Nk"])).astype(np.float32)
    return out


def synthetic_state_dict(config, seed=0, gain=1.0):
    """Seeded random weights of the right architecture (for timing runs without a checkpoint).
    torch.nn.Linear's default init range: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight and bias."""
    rng = np.random.default_rng(seed)
    sd = {}
    for key, shape in blob_schema(config):
        fan_in = shape[1] if len(shape) == 2 else None
        if fan_in is None:  # bias follows its weight's fan_in
            fan_in = sd[key.replace(".bias", ".weight")].shape[1]
        b = gain / np.sqrt(fan_in)
        sd[key] = rng.uniform(-b, b, size=shape).astype(np.float32)
    return unflatten_blob(config, flatten_state_dict(config, sd))


def stack_layers(state_dict_16, config_32, residual_scale=0.5):
    """Build weights for a 2x deeper ``sum`` stack by duplicating every layer of a trained model and
    scaling each copy's residual branch (last Linear of qpm, and ppm) by ``residual_scale``.

    Used to exercise the 32-layer i_v4_1 ARCHITECTURE with realistic activations: its trained blob
    is absent upstream (.MISSING_LARGE_BLOBS:16-17) while i_v4_0's (16 layers, same widths) is present.
    """
    c = normalise(config_32)
    out = {}
    for key, _ in blob_schema(c):
        if key.startswith("sum."):
            parts = key.split(".")
            src = ".".join(["sum", str(int(parts[1]) // 2)] + parts[2:])
            a = _to_numpy(state_dict_16[src]).astype(np.float32)
            if key.endswith("su.qpm.4.weight") or key.endswith("su.qpm.4.bias") or key.endswith("su.ppm.0.weight"):
                a = a * np.float32(residual_scale)
            out[key] = a
        else:
            out[key] = _to_numpy(state_dict_16[key]).astype(np.float32)
    return unflatten_blob(c, flatten_state_dict(c, out))
