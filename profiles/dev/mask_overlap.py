"""Could the dense-mask pass (49.5 us, HBM-read-bound) hide under the small kernels at the front of the forward if it ran on a second stream?
Upper bound of the idea: the pass on a side stream with NO data dependence on the forward of the same step (the forward consumes the
segments of the previous step) against the shipped order (same stream, mask pass first).   python profiles/dev/mask_overlap.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from pesto_amd import Model, CONFIGS

dev = torch.device("cuda:0")
cfg = CONFIGS["i_v4_1"]
sd, _ = bench.load_weights(cfg)
m = Model(cfg, validate=False, precision="f16_split").to(dev)
m.load_state_dict(sd)
X, ids, q, roa, R = bench.make_batch(3000, 8, 1, 30)
Xd, idsd, qd, road = [torch.from_numpy(v).to(dev) for v in (X, ids, q, roa)]
Md = torch.zeros((Xd.shape[0], R), dtype=torch.float32, device=dev)
Md[torch.arange(Xd.shape[0], device=dev), road.long()] = 1.0
main = torch.cuda.current_stream()
side = torch.cuda.Stream()

def serial():
    r, rr = m._segments(Md)
    return m.forward_segments(Xd, idsd, qd, r, rr)

prev = [m._segments(Md)]
def overlapped():
    side.wait_stream(main)
    with torch.cuda.stream(side):
        nxt = m._segments(Md)
    z = m.forward_segments(Xd, idsd, qd, prev[0][0], prev[0][1])
    main.wait_stream(side)
    prev[0] = nxt
    return z

def segs_only():
    return m.forward_segments(Xd, idsd, qd, road, R)

def timed(fn, n=40):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

for rep in range(3):
    print(f"rep {rep}: forward_segments only {timed(segs_only):.4f} ms | mask pass first, same stream {timed(serial):.4f} ms | mask pass on a side stream {timed(overlapped):.4f} ms", flush=True)
