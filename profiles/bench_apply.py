#!/usr/bin/env python3
"""The whole apply_model chain per structure, natively (SURVEY 8f rows 1-4 + the forward pass):
PDB text -> read / clean / split / filter / concatenate -> encode -> GPU k-NN -> forward (i_v4_1, stacked weights) -> GPU sigmoid +
b-factor expansion -> five b-factor PDB files. One structure at a time (latency), wall clock including host <-> device copies.
Reference's own figures for the same steps (profiling_analysis notebook, CUDA run): load 53 ms, process 68 ms, run 60 ms.
usage (GPU box): python profiles/bench_apply.py"""
import gzip
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import load_weights  # noqa: E402
from pesto_amd import Model  # noqa: E402
from pesto_amd.config import CONFIGS  # noqa: E402
from pesto_amd.structure_io import Structure  # noqa: E402

cfg = CONFIGS["i_v4_1"]
m = Model(cfg, validate=False).to("cuda")
m.load_state_dict(load_weights(cfg)[0])
tmp = tempfile.mkdtemp()
for name in ("7KHT_lipid", "1thf_D", "6I9F"):
    text = gzip.open(os.path.join(ROOT, "tests", "golden", "pdb", name + ".pdb.gz"), "rt").read()
    path = os.path.join(tmp, name + ".pdb")
    open(path, "w").write(text)
    stages = {k: [] for k in ("read", "preprocess", "encode", "knn+forward+post", "write x5", "total")}
    for it in range(12):
        t = [time.perf_counter()]
        s = Structure.read_pdb(path); t.append(time.perf_counter())
        s.preprocess(); t.append(time.perf_counter())
        X, q, roa, R = s.encode(30); t.append(time.perf_counter())
        Xd = torch.from_numpy(X).cuda()
        ids = m.knn_collate(Xd, [len(s)])
        z = m.forward_segments(Xd, ids, torch.from_numpy(q).cuda(), torch.from_numpy(roa).cuda(), R)
        p, bf = m.postprocess(z, torch.from_numpy(roa).cuda())
        bf = bf.cpu().numpy(); t.append(time.perf_counter())
        for c in range(bf.shape[0]):
            s.save_pdb(os.path.join(tmp, f"{name}_i{c}.pdb"), bf[c])
        t.append(time.perf_counter())
        if it >= 2:
            for k, a, b in zip(list(stages)[:5], t[:-1], t[1:]):
                stages[k].append((b - a) * 1e3)
            stages["total"].append((t[-1] - t[0]) * 1e3)
    print(json.dumps({"structure": name, "atoms_read": int(text.count("\nATOM") + text.count("\nHETATM") + 1), "atoms_kept": len(s), "residues": R,
                      "median_ms": {k: round(float(np.median(v)), 3) for k, v in stages.items()}}))
