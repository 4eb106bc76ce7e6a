"""Odd launch shapes through both kernel modes (rendezvous / node-wave, full and small launches): finite, deterministic, and
batch == one call per structure bit for bit.  GPU box: python profiles/stress_shapes.py"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from bench import load_weights  # noqa: E402
from pesto_amd import Model  # noqa: E402
from pesto_amd.config import CONFIGS  # noqa: E402

cfg = CONFIGS["i_v4_1"]
m = Model(cfg, precision="f16_split")
m.load_state_dict(load_weights(cfg)[0])
m.eval()
dev = torch.device("cuda:0")


def cloud(n, seed):
    rng = np.random.default_rng(seed)
    side = (n / 0.05) ** (1.0 / 3.0)
    X = (rng.random((n, 3)) * side).astype(np.float32)
    q = np.zeros((n, 30), np.float32)
    q[np.arange(n), rng.integers(0, 30, n)] = 1.0
    return X, q, (np.arange(n) // 8).astype(np.int32), (n + 7) // 8


for atoms, batch in ((100000, 1), (3000, 64), (2047, 3), (4097, 1), (6145, 2), (1025, 9), (20000, 2)):
    parts = [cloud(atoms, 11 * b + atoms) for b in range(batch)]
    X = torch.from_numpy(np.concatenate([p[0] for p in parts])).to(dev)
    q = torch.from_numpy(np.concatenate([p[1] for p in parts])).to(dev)
    r = parts[0][3]
    roa = torch.from_numpy(np.concatenate([p[2] + b * r for b, p in enumerate(parts)])).to(dev)
    sizes = [atoms] * batch
    ids = m.knn_collate(X, sizes)
    t0 = time.time()
    z1 = m.forward_segments(X, ids, q, roa, r * batch, sizes=sizes).clone()
    z2 = m.forward_segments(X, ids, q, roa, r * batch, sizes=sizes).clone()
    torch.cuda.synchronize()
    line = f"{batch} x {atoms}: finite {bool(torch.isfinite(z1).all())} deterministic {bool(torch.equal(z1, z2))} |z|max {float(z1.abs().max()):.2f} ({time.time() - t0:.2f} s)"
    if batch > 1:
        ok = True
        for b in (0, batch - 1):
            sl = slice(b * atoms, (b + 1) * atoms)
            idb = m.knn_collate(X[sl].contiguous(), [atoms])
            zb = m.forward_segments(X[sl].contiguous(), idb, q[sl].contiguous(), (roa[sl] - b * r).contiguous(), r)
            ok = ok and bool(torch.equal(zb, z1[b * r:(b + 1) * r]))
        line += f"  batch == singles bitwise: {ok}"
    print(line, flush=True)
