#!/usr/bin/env python3
"""Sweep (any host; minutes of CPU): the C oracle against the reference's logits on EVERY structure of the two large fixtures - the 53
pdbs_test chains x three models (cfg4_all53.npz) and the seven examples/ complexes x two trained models (examples_complexes.npz) -
fed the reference's own neighbour ids. The CPU suite runs a handful of these; this prints all of them.
Usage: python tests/golden/sweep_oracle.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from conftest import cfg4_all53, example_complexes, weights  # noqa: E402
from oracle import oracle  # noqa: E402
from pesto_amd.config import CONFIGS  # noqa: E402


def main():
    worst = {}
    models = {t: oracle.OracleModel(CONFIGS[t], weights(t)) for t in ("i_v4_1", "i_v4_0", "i_v3_0")}
    sets = [("pdbs_test", cfg4_all53(), (("i_v4_1", "z", "q0"), ("i_v4_0", "z_i_v4_0", "q0"), ("i_v3_0", "z_i_v3_0", "q0_all"))),
            ("examples", example_complexes(), (("i_v4_0", "z_i_v4_0", "q0"), ("i_v3_0", "z_i_v3_0", "q0_all")))]
    for set_name, items, combos in sets:
        for ch in items:
            errs = []
            for tag, zk, qk in combos:
                z = models[tag].forward_segments(ch["X"], ch["ids0"] + 1, ch[qk], ch["res_of_atom"], ch["R"])
                e = float(np.abs(z - ch[zk]).max())
                errs.append(f"{tag} {e:.2e}")
                worst[(set_name, tag)] = max(worst.get((set_name, tag), 0.0), e)
            print(f"{set_name:9s} {ch['name']:16s} N={ch['X'].shape[0]:6d}  " + "  ".join(errs), flush=True)
    for k, v in worst.items():
        print(f"max |oracle - reference| {k[0]} {k[1]}: {v:.2e}")
    return 0 if all(v < 1e-4 for v in worst.values()) else 1


if __name__ == "__main__":
    sys.exit(main())
