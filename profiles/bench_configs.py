#!/usr/bin/env python3
"""SURVEY 8(d) configs beside the headline one: per-step HIP-event times (median, p10, p90) with inputs resident in HBM.
  config 1  i_v4_1 (stacked weights), one real-size chain: N=2,810 R=355 (the 2AYO fixture's inputs)
  config 2  i_v4_1, 8 x synthetic N=3000 (the bench.py workload, repeated here for the percentiles)
  config 3  i_v3_0 (16 layers, N0 = 123, real weights), 8 x synthetic N=3000
  config 4  i_v4_1, the 53 chains of the reference's pdbs_test/ (N 1,641-3,052; coordinates from the parity fixture), batched to
            <= 24.6k atoms per launch;
            topology built on the GPU (knn_collate) inside the timed region
  config 5  i_v4_1, one synthetic N=20,000 structure, R=2,500
usage (GPU box): python profiles/bench_configs.py   -> one JSON line per config"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from bench import load_weights, make_batch  # noqa: E402
from conftest import golden, onehot, weights  # noqa: E402
from pesto_amd import Model  # noqa: E402
from pesto_amd.config import CONFIGS  # noqa: E402
from pesto_amd.sharding import batches  # noqa: E402
from pesto_amd.topology import synthetic_structure  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, n_struct, name, warm=5, reps=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ms.append(a.elapsed_time(b))
    ms = np.sort(np.array(ms))
    med, p10, p90 = float(np.median(ms)), float(np.percentile(ms, 10)), float(np.percentile(ms, 90))
    print(json.dumps({"config": name, "structures_per_step": n_struct, "ms_per_step": {"median": med, "p10": p10, "p90": p90},
                      "structures_per_s": n_struct / med * 1e3, "ms_per_structure": med / n_struct}), flush=True)


def dev_args(X, ids, q, roa):
    return [torch.from_numpy(np.ascontiguousarray(v)).to(dev) for v in (X, ids, q, roa)]


def model(tag, sd):
    m = Model(CONFIGS[tag], validate=False).to(dev)
    m.load_state_dict(sd)
    return m


m41 = model("i_v4_1", load_weights(CONFIGS["i_v4_1"])[0])

g = golden("fwd_i_v4_1_stacked_2AYO")
g0 = golden("fwd_i_v4_0_2AYO")        # the inputs live with the i_v4_0 fixture of the same chain
a = dev_args(g0["X"], g0["ids_topk"].astype(np.int64), onehot(g0["q_idx"], 30), g0["res_of_atom"])
R1 = int(g["z"].shape[0])
timed(lambda: m41.forward_segments(a[0], a[1], a[2], a[3], R1), 1, "1: i_v4_1, one chain N=2810 R=355 (2AYO inputs)")
z = m41.forward_segments(a[0], a[1], a[2], a[3], R1).cpu().numpy()
assert np.abs(z - g["z"]).max() < 1e-4

X, ids, q, roa, R = make_batch(3000, 8, 1, 30)
a2 = dev_args(X, ids, q, roa)
timed(lambda: m41.forward_segments(a2[0], a2[1], a2[2], a2[3], R), 8, "2: i_v4_1, 8 x N=3000 (headline)")

m30 = model("i_v3_0", weights("i_v3_0"))
X, ids, q, roa, R3 = make_batch(3000, 8, 1, 123)
a3 = dev_args(X, ids, q, roa)
timed(lambda: m30.forward_segments(a3[0], a3[1], a3[2], a3[3], R3), 8, "3: i_v3_0 (16 layers, N0=123), 8 x N=3000")

from bench import config4_structures  # noqa: E402
chains, sizes, _ = config4_structures(53, m41, "dense")        # the 53 REAL pdbs_test chains (tests/golden/cfg4_all53.npz)
groups = batches(list(range(53)), sizes, 24600)
prepared = []
for grp in groups:
    Xg = np.concatenate([chains[i][0] for i in grp])
    qg = np.concatenate([chains[i][2] for i in grp])
    roas, r0 = [], 0
    for i in grp:
        M = chains[i][3]
        roas.append(M.argmax(1).astype(np.int32) + r0)
        r0 += M.shape[1]
    prepared.append((torch.from_numpy(Xg).to(dev), torch.from_numpy(qg).to(dev), torch.from_numpy(np.concatenate(roas)).to(dev), r0,
                     [sizes[i] for i in grp]))


def run4():
    for Xd, qd, rd, Rg, sz in prepared:
        ids_d = m41.knn_collate(Xd, sz)
        m41.forward_segments(Xd, ids_d, qd, rd, Rg)


timed(run4, 53, f"4: i_v4_1, the 53 pdbs_test chains N 1641-3052 ({sum(sizes)} atoms) in {len(groups)} launches, GPU k-NN included", warm=2, reps=10)

X, ids, q, roa, R5 = make_batch(20000, 1, 1, 30)
a5 = dev_args(X, ids, q, roa)
timed(lambda: m41.forward_segments(a5[0], a5[1], a5[2], a5[3], R5), 1, "5: i_v4_1, one structure N=20000 R=2500", warm=3, reps=10)
