"""Experiment: do two forwards on two HIP streams (two handles, two workspaces) overlap each other's launch gaps / prologues / tails?
python profiles/dev/two_streams.py [precision]"""
import sys, os, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from pesto_amd import Model
from pesto_amd.config import CONFIGS
prec = sys.argv[1] if len(sys.argv) > 1 else "f16_split"
cfg = CONFIGS["i_v4_1"]
sd, _ = bench.load_weights(cfg)
dev = torch.device("cuda:0")
def mk(seed):
    m = Model(cfg, validate=False, precision=prec, async_auto=True).to(dev); m.load_state_dict(sd)
    X, ids, q, roa, R = bench.make_batch(3000, 8, seed, 30)
    return m, [torch.from_numpy(a).to(dev) for a in (X, ids, q, roa)], R
A, B = mk(1), mk(1001)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
def run(n, two):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        m, a, R = (A, B)[i & 1] if two else A
        with torch.cuda.stream((sA, sB)[i & 1] if two else sA):
            m.forward_segments(*a, R)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for two in (False, True, False, True):
    run(6, two)
    print("two streams" if two else "one stream ", "%.3f ms per step  %.1f structures/s" % ((lambda t: (t, 8e3 / t))(run(40, two))), flush=True)
