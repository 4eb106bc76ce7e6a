"""Atom order and the gathers: one N = 20,000 structure (SURVEY config 5) and the headline batch, atoms in generation order (no spatial
locality; what bench.py uses) and along a Z-order curve (chain-like locality, as PDB files have).   python profiles/dev/order_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from pesto_amd import Model, CONFIGS

dev = torch.device("cuda:0")
cfg = CONFIGS["i_v4_1"]
sd, _ = bench.load_weights(cfg)
m = Model(cfg, validate=False, precision="f16_split").to(dev)
m.load_state_dict(sd)
for n, b in ((20000, 1), (3000, 8)):
    for order in ("random", "morton"):
        X, ids, q, roa, R = bench.make_batch(n, b, 1, 30, order)
        a = [torch.from_numpy(v).to(dev) for v in (X, ids, q, roa)] + [R]
        for _ in range(8): m.forward_segments(*a)
        torch.cuda.synchronize()
        ms = []
        for _ in range(20):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); m.forward_segments(*a); e1.record(); e1.synchronize()
            ms.append(e0.elapsed_time(e1))
        print(f"{b} x N = {n:6d}  order {order:7s}: median {np.median(ms):.3f} ms  ({n * b / np.median(ms) / 1e3:.2f} M atoms/s)", flush=True)
