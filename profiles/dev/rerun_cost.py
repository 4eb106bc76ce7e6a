"""What AUTO's fp32 repeat costs when ONE small structure of a full launch is flagged: 8 structures of 3,000 atoms + one 40-atom structure
(zero-padded neighbour slots: the pad trigger flags it) in one collated launch; against the same launch without it, and on the exact
kernels as a whole.   python profiles/dev/rerun_cost.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from pesto_amd import Model, CONFIGS
from pesto_amd.topology import synthetic_structure

dev = torch.device("cuda:0")
cfg = CONFIGS["i_v4_1"]
sd, _ = bench.load_weights(cfg)
m = Model(cfg, validate=False, precision="auto").to(dev)
m.load_state_dict(sd)
big = [list(synthetic_structure(3000, 10 + b, n0=30)) for b in range(8)]
small = list(synthetic_structure(40, 99, n0=30))
def dev_struct(s): return [torch.as_tensor(np.asarray(v)).to(dev) for v in s]
def timed(structs, n=10):
    for _ in range(3): z = m.forward_batch(structs, independent=True)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): z = m.forward_batch(structs, independent=True)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3, z
for rep in range(2):
    m.set_precision("auto"); n0 = m.status()["n_fp32_rerun"]
    t_clean, _ = timed(big)
    t_mixed, zm = timed(big + [small]); n1 = m.status()["n_fp32_rerun"]
    m.set_precision("fp32"); t_fp32, zf = timed(big + [small])
    m.set_precision("f16_split"); t_split, zs = timed(big + [small])
    ok_small = bool(np.array_equal(np.asarray(zm[-1]), np.asarray(zf[-1])))
    ok_big = all(np.array_equal(np.asarray(zm[i]), np.asarray(zs[i])) for i in range(8))
    print(f"rep {rep}: 8 x 3000 auto {t_clean:.3f} ms | + one 40-atom structure: auto {t_mixed:.3f} ms ({(n1 - n0) / 13:.0f} repeat per call), f16_split {t_split:.3f}, "
          f"fp32 {t_fp32:.3f} | flagged structure == fp32 bits: {ok_small}, the others == f16_split bits: {ok_big}", flush=True)
