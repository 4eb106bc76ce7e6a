#!/usr/bin/env python3
"""Timing of the trajectory mode and the device post-op (row 4): i_v4_1, one synthetic N=3000 structure,
F frames with frame-0 topology. Compares Model.forward_frames (several frames per launch) with the reference's call pattern,
one Model.forward per frame (md_analysis/apply_model_md.ipynb cell 6). Run on the GPU box:  python profiles/bench_frames.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import load_weights, make_batch  # noqa: E402
from pesto_amd import Model  # noqa: E402
from pesto_amd.config import CONFIGS  # noqa: E402

cfg = CONFIGS["i_v4_1"]
sd, _ = load_weights(cfg)
m = Model(cfg, validate=False).to("cuda")
m.load_state_dict(sd)
N, F = 3000, 64
X, ids, q, roa, R = make_batch(N, 1, 1, 30)
rng = np.random.default_rng(0)
Xf = np.stack([X] + [(X + rng.normal(0, 0.3, X.shape)).astype(np.float32) for _ in range(F - 1)], 1)   # [N, F, 3]
Xd = torch.from_numpy(Xf).cuda()
idd, qd, rd = torch.from_numpy(ids).cuda(), torch.from_numpy(q).cuda(), torch.from_numpy(roa).cuda()


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out


t_loop, z_loop = timed(lambda: torch.stack([m.forward_segments(Xd[:, f].contiguous(), idd, qd, rd, R) for f in range(F)]))
print(f"per-frame calls (reference pattern): {t_loop / F * 1e3:.3f} ms/frame  ({F / t_loop:.0f} frames/s)")
for fpl in (0, 4, 8, 16):
    t, z = timed(lambda: m.forward_frames_segments(Xd, idd, qd, rd, R, frame_axis=1, frames_per_launch=fpl))
    assert torch.equal(z, z_loop)
    print(f"forward_frames, frames_per_launch={fpl or 'auto'}: {t / F * 1e3:.3f} ms/frame  ({F / t:.0f} frames/s)  x{t_loop / t:.2f}")
t, z = timed(lambda: m.forward_frames_segments(Xf, ids, q, roa, R, frame_axis=1))
print(f"forward_frames from HOST numpy [N,F,3] (pack + H2D + D2H included): {t / F * 1e3:.3f} ms/frame  ({F / t:.0f} frames/s)")
t, _ = timed(lambda: m.postprocess(z_loop[0], rd), reps=50)
print(f"postprocess (sigmoid + expansion to {N} atoms x 5 channels), device tensors: {t * 1e6:.1f} us per structure")
