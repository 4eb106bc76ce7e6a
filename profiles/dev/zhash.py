"""sha256 of the logits of the bench workload (8 x 3,000 atoms, f16_split and fp32) and of one real-size chain under the library PESTO_LIB names:
two builds that print the same lines compute the same bits.   PESTO_LIB=... python profiles/dev/zhash.py"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from pesto_amd import Model, CONFIGS

dev = torch.device("cuda:0")
cfg = CONFIGS["i_v4_1"]
sd, _ = bench.load_weights(cfg)
for prec in ("f16_split", "fp32"):
    m = Model(cfg, validate=False, precision=prec).to(dev)
    m.load_state_dict(sd)
    for n, b in ((3000, 8), (2810, 1), (20000, 1)):
        X, ids, q, roa, R = bench.make_batch(n, b, 1, 30)
        a = [torch.from_numpy(v).to(dev) for v in (X, ids, q, roa)] + [R]
        z = m.forward_segments(*a).cpu().numpy()
        print(prec, f"{b} x {n}", hashlib.sha256(z.tobytes()).hexdigest()[:16], float(np.abs(z).max()), flush=True)
