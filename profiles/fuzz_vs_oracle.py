#!/usr/bin/env python3
"""Randomised parity sweep on the GPU box: ragged batches of random structures (sizes from 2 atoms to a few thousand, including the
N < 64 / N = 64 / 65 edges, residues of 1 - 30 atoms, non-contiguous residues, k < 64 neighbour tables) through pesto_forward_batch in
both batch modes and through the pipelined submit / wait path (with its byte-index feature upload) (one-atom structures are left out: alone, such a "structure" has max(D) = 0
and the reference itself divides 0 by 0, src/model_operations.py:12-20), against the C oracle (per-structure calls for INDEPENDENT, the collated
call for COLLATED) and bitwise against one-call-per-structure. Trained i_v4_0 weights by default. usage: python profiles/fuzz_vs_oracle.py [rounds] [i_v4_0 | i_v3_0 | i_v4_1]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import weights  # noqa: E402
from oracle import oracle  # noqa: E402
from pesto_amd import Model  # noqa: E402
from pesto_amd.config import CONFIGS  # noqa: E402
from pesto_amd.topology import collate_batch_features, extract_topology, mask_to_segments, synthetic_cloud  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
TAG = sys.argv[2] if len(sys.argv) > 2 else "i_v4_0"       # i_v4_0 | i_v3_0 (123 input features) | i_v4_1 (32 layers, stacked weights)
rng = np.random.default_rng(2024)
cfg = CONFIGS[TAG]
N0 = cfg["em"]["N0"]
m = Model(cfg)
m.load_state_dict(weights(TAG))
o = oracle.OracleModel(cfg, weights(TAG), wide=True)      # double accumulators, float32 storage: within 5e-5 of the fp64 reference where the
# float32-accumulating port is 7e-4 off (tests/test_fuzz_pins.py)
# random clouds are harsher than proteins (|z| up to 15, 2-atom members, states |p| ~ 50): the REFERENCE's own fp32 run is up to 1.2e-4
# from its fp64 run on them (tests/golden/make_fuzz_pins.py), so the sweep asserts 2.5e-4 against the wide oracle (no factor on top);
# the pinned cases and the seeded leg with the reference's own logits are tests/test_fuzz_pins.py
TOL = 2.5e-4
SIZES = [2, 3, 17, 40, 63, 64, 65, 66, 100, 129, 500, 1023, 1024, 1025, 2000, 3100]


def structure(n, seed):
    X = synthetic_cloud(n, seed)
    k = min(64, n)
    ids = np.asarray(extract_topology(X, 64)).astype(np.int32)
    if n > 80 and seed % 3 == 0:      # a shorter neighbour table (k < 64 columns)
        k = int(rng.choice([8, 16, 33]))
        ids = np.ascontiguousarray(ids[:, :k])
    q = np.zeros((n, N0), np.float32)
    for lo_, hi_ in ((0, 30),) if N0 == 30 else ((0, 30), (30, 59), (59, 123)):      # one-hot per feature block (element | residue | atom name)
        q[np.arange(n), rng.integers(lo_, hi_, n)] = 1.0
    # residues: runs of 1 - 30 atoms; sometimes the residue ids are permuted so that columns are not in atom order
    cuts, i = [], 0
    while i < n:
        i += int(rng.integers(1, 31)); cuts.append(min(i, n))
    roa = np.zeros(n, np.int64); a = 0
    for r, b in enumerate(cuts):
        roa[a:b] = r; a = b
    R = len(cuts)
    if seed % 4 == 1:
        roa = rng.permutation(R)[roa]
    M = np.zeros((n, R), np.float32); M[np.arange(n), roa] = 1.0
    return X, ids, q, M


worst = 0.0
t0 = time.time()
for it in range(rounds):
    nb = int(rng.integers(1, 9))
    sizes = [int(rng.choice(SIZES)) for _ in range(nb)]
    structs = [structure(n, 1000 * it + j) for j, n in enumerate(sizes)]
    # INDEPENDENT: every member as in its own call
    z_ind = m.forward_batch(structs, independent=True)
    z_pipe = m.forward_batch_wait(m.forward_batch_submit(structs, independent=True))
    for j, st in enumerate(structs):
        roa, R = mask_to_segments(st[3])
        ids_pad = np.zeros((st[0].shape[0], 64), np.int32)          # what collate_batch_features hands the forward: 1-based, zero-padded to 64
        ids_pad[:, :st[1].shape[1]] = st[1] + 1
        z_ref = o.forward_segments(st[0], ids_pad, st[2], roa, R)
        single = m.forward_batch([st], independent=True)[0]
        e = float(np.abs(z_ind[j] - z_ref).max())
        worst = max(worst, e)
        if not e < 1e-4:
            m.set_precision("fp32")
            z32 = m.forward_batch([st], independent=True)[0]
            m.set_precision("auto")
            print(f"   independent round {it} structure {j} ({sizes[j]} atoms, k = {st[1].shape[1]}): max err {e:.2e}, |z|max {np.abs(z_ref).max():.1f}; "
                  f"exact-fp32 kernels vs oracle {np.abs(z32 - z_ref).max():.2e}", flush=True)
        assert e < TOL, ("independent", it, j, sizes[j], e)
        assert np.array_equal(z_ind[j], single) and np.array_equal(z_pipe[j], single), ("bitwise", it, j, sizes[j])
    # COLLATED: the reference's forward on the collated batch (wrap target = last atom of the batch, one max(D))
    Xc, idc, qc, Mc = collate_batch_features([list(s) for s in structs])
    roa_c, R_c = mask_to_segments(Mc)
    z_col = np.concatenate(m.forward_batch(structs, independent=False), 0)
    z_ref = o.forward_segments(Xc, idc, qc, roa_c, R_c)
    e = float(np.abs(z_col - z_ref).max())
    if not e < 1e-4:      # where, and is it the split arithmetic or fp32 re-association? (the exact-fp32 kernels on the same batch)
        err = np.abs(z_col - z_ref)
        r = int(err.max(1).argmax())
        offs = np.cumsum([0] + [s_[3].shape[1] for s_ in structs])
        owner = int(np.searchsorted(offs, r, side="right") - 1)
        m.set_precision("fp32")
        z32 = np.concatenate(m.forward_batch(structs, independent=False), 0)
        m.set_precision("auto")
        print(f"   collated round {it}: max err {e:.2e} at residue {r} (structure {owner}, {sizes[owner]} atoms), z_ref {z_ref[r]}, hip {z_col[r]}; "
              f"exact-fp32 kernels vs oracle {np.abs(z32 - z_ref).max():.2e}, per structure " +
              " ".join(f"{np.abs(z_col[offs[j]:offs[j + 1]] - z_ref[offs[j]:offs[j + 1]]).max():.1e}" for j in range(len(structs))), flush=True)
    worst = max(worst, e)
    assert e < TOL, ("collated", it, sizes, e)
    print(f"round {it}: sizes {sizes}  max |hip - oracle| so far {worst:.2e}", flush=True)
print(f"{TAG}: {rounds} rounds ok in {time.time() - t0:.0f} s, max |hip - oracle| = {worst:.2e}; fp32 re-runs {m.status()['n_fp32_rerun']}")
