"""One structure per call after the GPU has been idle: groups of 10 calls, back to back, after 0.5 s of host-only time - how long the
first calls of a burst are slower than the steady state (clock ramp of an idle GPU).   python profiles/dev/lat_ramp.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from pesto_amd import Model, CONFIGS

dev = torch.device("cuda:0")
cfg = CONFIGS["i_v4_1"]
sd, _ = bench.load_weights(cfg)
m = Model(cfg, validate=False, precision="auto").to(dev)
m.load_state_dict(sd)
X, ids, q, roa, R = bench.make_batch(3000, 1, 1, 30)
a = [torch.from_numpy(v).to(dev) for v in (X, ids, q, roa)] + [R]
for _ in range(3): m.forward_segments(*a)
torch.cuda.synchronize()
for idle in (0.5, 0.05, 2.0):
    for rep in range(2):
        time.sleep(idle)
        row = []
        for grp in range(8):
            t = time.perf_counter()
            for _ in range(10): m.forward_segments(*a)
            torch.cuda.synchronize(); row.append(round((time.perf_counter() - t) / 10 * 1e3, 4))
        print(f"idle {idle} s, groups of 10 calls (ms per call):", row, flush=True)
