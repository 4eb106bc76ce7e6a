#!/usr/bin/env python3
"""developer: where the main thread of apply_model spends its time (192 files, i_v4_1): waiting for the loader threads, packing a
launch, queueing GPU work, waiting for the GPU. usage (GPU box): python profiles/dev/bulk_breakdown.py"""
import gzip, os, sys, tempfile, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from concurrent.futures import ThreadPoolExecutor
from bench import load_weights
from pesto_amd import Model
from pesto_amd.apply import _load
from pesto_amd.config import CONFIGS
cfg = CONFIGS["i_v4_1"]
m = Model(cfg, validate=False).to("cuda")
m.load_state_dict(load_weights(cfg)[0])
dev = torch.device("cuda", 0)
tmp = tempfile.mkdtemp()
texts = [gzip.open(os.path.join(ROOT, "tests", "golden", "pdb", n + ".pdb.gz"), "rt").read() for n in ("7KHT_lipid", "1thf_D", "6I9F")]
paths = []
for i in range(192):
    p = os.path.join(tmp, f"s{i:03d}.pdb")
    open(p, "w").write(texts[i % 3]); paths.append(p)
for rep in range(2):
    T = dict(load_wait=0.0, pack=0.0, queue=0.0, gpu_wait=0.0, slice=0.0)
    t_all = time.perf_counter()
    with ThreadPoolExecutor(max_workers=8) as pool:
        loads = [pool.submit(_load, p, 30) for p in paths]
        group, atoms = [], 0
        def flush(group):
            t0 = time.perf_counter()
            sizes = [len(g[0]) for g in group]
            Xh = np.concatenate([g[1] for g in group]); qh = np.concatenate([g[2] for g in group])
            r_off = np.cumsum([0] + [g[4] for g in group])
            rh = np.concatenate([g[3] + r_off[i] for i, g in enumerate(group)]).astype(np.int32)
            t1 = time.perf_counter()
            X = torch.from_numpy(Xh).to(dev); q = torch.from_numpy(qh).to(dev); roa = torch.from_numpy(rh).to(dev)
            ids = m.knn_collate(X, sizes)
            z = m.forward_segments(X, ids, q, roa, int(r_off[-1]), sizes=sizes)
            p, bf = m.postprocess(z, roa)
            t2 = time.perf_counter()
            p, bf = p.cpu().numpy(), bf.cpu().numpy()
            t3 = time.perf_counter()
            T["pack"] += t1 - t0; T["queue"] += t2 - t1; T["gpu_wait"] += t3 - t2
        for fut in loads:
            t0 = time.perf_counter()
            s, X, q, roa, R = fut.result()
            T["load_wait"] += time.perf_counter() - t0
            if group and atoms + len(s) > 24576:
                flush(group); group, atoms = [], 0
            group.append((s, X, q, roa, R)); atoms += len(s)
        flush(group)
    tot = time.perf_counter() - t_all
    print(f"total {tot*1e3:.1f} ms for 192 files = {192/tot:.0f} /s | " + " ".join(f"{k} {v*1e3:.1f} ms" for k, v in T.items()))
t0 = time.perf_counter()
for p in paths[:48]:
    _load(p, 30)
print(f"one thread: _load {1e3*(time.perf_counter()-t0)/48:.2f} ms per file")
