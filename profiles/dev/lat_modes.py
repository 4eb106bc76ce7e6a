"""One structure per call (3,000 atoms, device tensors): the default policy (auto: the call returns checked), auto with the check deferred
(pesto_set_async_auto) and f16_split without any check - what the per-call host wait costs.   python profiles/dev/lat_modes.py"""
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
def timed(n=50):
    for _ in range(5): m.forward_segments(*a)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): m.forward_segments(*a)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
for rep in range(3):
    row = {}
    m.set_precision("auto").set_async_auto(False); row["auto"] = timed()
    m.set_async_auto(True); row["auto, deferred check"] = timed(); m.synchronize(); m.set_async_auto(False)
    m.set_precision("f16_split"); row["f16_split (no check)"] = timed()
    print("rep", rep, {k: round(v, 4) for k, v in row.items()}, "env", {k: os.environ.get(k) for k in ("HIP_FORCE_DEV_KERNARG", "ROC_ACTIVE_WAIT_TIMEOUT")}, flush=True)
