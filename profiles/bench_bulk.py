#!/usr/bin/env python3
"""End-to-end bulk throughput of pesto_amd.apply.apply_model: N PDB files on disk -> probabilities (+ 5 b-factor PDB files each),
i_v4_1, host stages in a thread pool, ~24k atoms per GPU launch. usage (GPU box): python profiles/bench_bulk.py"""
import gzip, json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import load_weights
from pesto_amd import Model
from pesto_amd.apply import apply_model
from pesto_amd.config import CONFIGS
cfg = CONFIGS["i_v4_1"]
m = Model(cfg, validate=False).to("cuda")
m.load_state_dict(load_weights(cfg)[0])
tmp = tempfile.mkdtemp()
texts = [gzip.open(os.path.join(ROOT, "tests", "golden", "pdb", n + ".pdb.gz"), "rt").read() for n in ("7KHT_lipid", "1thf_D", "6I9F")]
paths = []
for i in range(192):
    p = os.path.join(tmp, f"s{i:03d}.pdb")
    open(p, "w").write(texts[i % 3])
    paths.append(p)
apply_model(m, paths[:24], write=True)        # warm-up (workspace, page cache)
for write in (False, True):
    for workers in (1, 8, 16):
        t0 = time.perf_counter()
        res = apply_model(m, paths, write=write, workers=workers)
        dt = time.perf_counter() - t0
        print(json.dumps({"files": len(res), "write_pdb": write, "workers": workers, "seconds": round(dt, 3), "structures_per_s": round(len(res) / dt, 1)}))
