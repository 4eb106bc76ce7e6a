#!/usr/bin/env python3
"""Pins for the randomised parity sweep (build container only: IMPORTS the reference from /root/reference).

Round 3's sweep on the GPU (profiles/fuzz_vs_oracle.py -> profiles/r03_fuzz.txt) found two inputs where |hip - oracle| exceeded the
north-star's 1e-4: the collated batch of its round 2 (it contains a 2-atom structure: 62 of its 64 neighbour slots wrap to the last
atom of the call, src/model_operations.py:8) and structure 6 of its round 18 (500 atoms, a neighbour table of k = 8 columns
zero-padded to 64, :230). This script REPLAYS the sweep's generator (same seeds, same draw order, no GPU needed), recovers those
two inputs and runs the REFERENCE on them in fp32 with 1 and 8 threads and in fp64 - the reference's own spread is what decides
whether 3e-4 is a defect or noise. It also draws the batches of the seeded fuzz leg of `pytest -m gpu` (three rounds of ragged
batches) and stores the reference's fp32 / fp64 logits for every structure, so that the test can hold 1e-4 wherever the reference
itself is conditioned to 1e-5.

Output: tests/golden/fuzz_pins.npz (inputs, logits, spreads).   usage: python tests/golden/make_fuzz_pins.py
"""
import os
import sys

import numpy as np
import torch as pt

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from pesto_amd.topology import extract_topology, synthetic_cloud  # noqa: E402

SIZES = [2, 3, 17, 40, 63, 64, 65, 66, 100, 129, 500, 1023, 1024, 1025, 2000, 3100]
N0 = 30


def structure(rng, n, seed):
    """profiles/fuzz_vs_oracle.py::structure, draw for draw (q as indices, the mask as res_of_atom)."""
    X = synthetic_cloud(n, seed)
    ids = np.asarray(extract_topology(X, 64)).astype(np.int32)
    if n > 80 and seed % 3 == 0:
        k = int(rng.choice([8, 16, 33]))
        ids = np.ascontiguousarray(ids[:, :k])
    qi = rng.integers(0, 30, n)
    cuts, i = [], 0
    while i < n:
        i += int(rng.integers(1, 31)); cuts.append(min(i, n))
    roa = np.zeros(n, np.int64); a = 0
    for r, b in enumerate(cuts):
        roa[a:b] = r; a = b
    R = len(cuts)
    if seed % 4 == 1:
        roa = rng.permutation(R)[roa]
    return X, ids, qi.astype(np.int16), roa.astype(np.int32), R


def replay(rounds):
    rng = np.random.default_rng(2024)
    out = []
    for it in range(rounds):
        nb = int(rng.integers(1, 9))
        sizes = [int(rng.choice(SIZES)) for _ in range(nb)]
        out.append([structure(rng, n, 1000 * it + j) for j, n in enumerate(sizes)])
    return out


def collate_np(structs):
    """collate_batch_features (src/dataset.py:91-112) through the reference's own function."""
    items = []
    for X, ids, qi, roa, R in structs:
        q = np.zeros((X.shape[0], N0), np.float32); q[np.arange(X.shape[0]), qi] = 1.0
        M = np.zeros((X.shape[0], R), np.float32); M[np.arange(X.shape[0]), roa] = 1.0
        items.append([pt.from_numpy(X), pt.from_numpy(ids.astype(np.int64)), pt.from_numpy(q), pt.from_numpy(M)])
    return mg.collate(items)


def reference_runs(model, model64, Xc, idsc, qc, Mc):
    """the reference's logits: fp32 with 1 thread, fp32 with 8 threads, fp64"""
    out = {}
    for nt in (1, 8):
        pt.set_num_threads(nt)
        out[f"z32_t{nt}"] = mg.run_forward(model, Xc, idsc, qc, Mc)
    with pt.no_grad():
        out["z64"] = model64(Xc.double(), idsc, qc.double(), Mc.double()).numpy()
    return out


def main():
    import copy
    cfg, model = mg.load_run("i_v4_0_2021-09-07_11-20")
    model64 = copy.deepcopy(model).double()
    arrs = {}
    # ------------------------------------------------------------------ the two inputs of round 3's sweep
    print("replaying the sweep's generator (19 rounds) ...", flush=True)
    rounds = replay(19)
    # (a) collated round 2
    st = rounds[2]
    Xc, idsc, qc, Mc = collate_np(st)
    r = reference_runs(model, model64, Xc, idsc, qc, Mc)
    sizes = np.array([[s[0].shape[0], s[4]] for s in st], np.int32)
    print("collated round 2: sizes", sizes[:, 0].tolist())
    print("   reference fp32(1 thread) vs fp64 %.2e, fp32(8) vs fp64 %.2e, fp32 1 vs 8 threads %.2e, |z|max %.1f" % (
        np.abs(r["z32_t1"] - r["z64"]).max(), np.abs(r["z32_t8"] - r["z64"]).max(), np.abs(r["z32_t1"] - r["z32_t8"]).max(), np.abs(r["z64"]).max()))
    offs = np.cumsum([0] + sizes[:, 1].tolist())
    print("   per structure (fp32 t8 vs fp64):", " ".join("%.1e" % np.abs(r["z32_t8"][offs[j]:offs[j + 1]] - r["z64"][offs[j]:offs[j + 1]]).max() for j in range(len(st))))
    arrs.update(a_X=Xc.numpy(), a_ids=idsc.numpy().astype(np.int32), a_q=qc.numpy().argmax(1).astype(np.int16),
                a_roa=mg.res_of_atom(Mc), a_sizes=sizes, a_z32_t1=r["z32_t1"], a_z32_t8=r["z32_t8"], a_z64=r["z64"])
    # (b) independent round 18, structure 6
    st = [rounds[18][6]]
    Xc, idsc, qc, Mc = collate_np(st)
    r = reference_runs(model, model64, Xc, idsc, qc, Mc)
    print("independent round 18 structure 6: %d atoms, k = %d columns" % (st[0][0].shape[0], st[0][1].shape[1]))
    print("   reference fp32(1 thread) vs fp64 %.2e, fp32(8) vs fp64 %.2e, fp32 1 vs 8 threads %.2e, |z|max %.1f" % (
        np.abs(r["z32_t1"] - r["z64"]).max(), np.abs(r["z32_t8"] - r["z64"]).max(), np.abs(r["z32_t1"] - r["z32_t8"]).max(), np.abs(r["z64"]).max()))
    arrs.update(b_X=Xc.numpy(), b_ids=idsc.numpy().astype(np.int32), b_q=qc.numpy().argmax(1).astype(np.int16),
                b_roa=mg.res_of_atom(Mc), b_k=np.int32(st[0][1].shape[1]), b_z32_t1=r["z32_t1"], b_z32_t8=r["z32_t8"], b_z64=r["z64"])
    # ------------------------------------------------------------------ the seeded fuzz leg of pytest -m gpu: 3 rounds of ragged batches,
    # every structure alone (the reference's bulk loop) AND the collated batch; sizes capped so that the CPU reference stays in minutes
    rng = np.random.default_rng(4242)
    LEG_SIZES = [2, 3, 17, 40, 63, 64, 65, 66, 100, 129, 500, 1023, 1024, 1025]
    leg = []
    for it in range(3):
        nb = int(rng.integers(3, 8))
        sizes = [int(rng.choice(LEG_SIZES)) for _ in range(nb)]
        leg.append([structure(rng, n, 7000 + 100 * it + j) for j, n in enumerate(sizes)])
    for it, st in enumerate(leg):
        pre = f"leg{it}_"
        Xc, idsc, qc, Mc = collate_np(st)
        r = reference_runs(model, model64, Xc, idsc, qc, Mc)
        sizes = np.array([[s[0].shape[0], s[4]] for s in st], np.int32)
        ks = np.array([s[1].shape[1] for s in st], np.int32)
        # every member alone
        z1, z64 = [], []
        for s in st:
            a = collate_np([s])
            rr = reference_runs(model, model64, *a)
            z1.append(rr["z32_t8"]); z64.append(rr["z64"])
        z1 = np.concatenate(z1, 0); z64 = np.concatenate(z64, 0)
        offs = np.cumsum([0] + sizes[:, 1].tolist())
        print(f"leg round {it}: sizes {sizes[:, 0].tolist()} k {ks.tolist()}")
        print("   collated: fp32 vs fp64 per structure " + " ".join("%.1e" % np.abs(r["z32_t8"][offs[j]:offs[j + 1]] - r["z64"][offs[j]:offs[j + 1]]).max() for j in range(len(st))))
        print("   alone   : fp32 vs fp64 per structure " + " ".join("%.1e" % np.abs(z1[offs[j]:offs[j + 1]] - z64[offs[j]:offs[j + 1]]).max() for j in range(len(st))))
        arrs.update({pre + "X": Xc.numpy(), pre + "ids": idsc.numpy().astype(np.int32), pre + "q": qc.numpy().argmax(1).astype(np.int16),
                     pre + "roa": mg.res_of_atom(Mc), pre + "sizes": sizes, pre + "k": ks,
                     pre + "col_z32": r["z32_t8"], pre + "col_z64": r["z64"], pre + "ind_z32": z1, pre + "ind_z64": z64})
    mg.save("fuzz_pins", **arrs)


if __name__ == "__main__":
    main()
