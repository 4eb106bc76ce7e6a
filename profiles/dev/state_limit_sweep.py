"""Calibration of PESTO_AUTO_STATE_LIMIT_DEFAULT (GPU box): how many structures precision "auto" repeats on the exact kernels as a function of the
state limit, over everything the trigger must NOT fire on (53 pdbs_test chains x 3 models, 7 complexes x 2 models, the bench workload,
the MD frames) and the two pinned ill-conditioned inputs it exists for.   python profiles/dev/state_limit_sweep.py > gpurun_out/state_limit.txt"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import cfg4_all53, example_complexes, golden, onehot, weights  # noqa: E402
from pesto_amd import Model  # noqa: E402
from pesto_amd.config import CONFIGS  # noqa: E402
from pesto_amd.topology import mask_to_segments, synthetic_structure  # noqa: E402
import test_fuzz_pins as fp  # noqa: E402


def dense_mask(roa, R):
    M = np.zeros((roa.size, R), np.float32); M[np.arange(roa.size), roa] = 1.0
    return M


def datasets():
    chains = cfg4_all53()
    cx = example_complexes()
    out = []
    for tag, qkey in (("i_v4_1", "q0"), ("i_v4_0", "q0"), ("i_v3_0", "q0_all")):
        out.append((f"53 pdbs_test chains, {tag}", tag, [(c["X"], c["ids0"], c[qkey], dense_mask(c["res_of_atom"], c["R"])) for c in chains]))
    for tag, qkey in (("i_v4_0", "q0"), ("i_v3_0", "q0_all")):
        out.append((f"7 example complexes, {tag}", tag, [(c["X"], c["ids0"], c[qkey], dense_mask(c["res_of_atom"], c["R"])) for c in cx]))
    synth = [synthetic_structure(3000, 1 + b, n0=30) for b in range(8)]
    out.append(("bench workload (8 x synthetic N=3000), i_v4_1", "i_v4_1", [(X, ids0, q, M) for X, ids0, q, M in synth]))
    synth3 = [synthetic_structure(3000, 1 + b, n0=123) for b in range(4)]
    out.append(("4 x synthetic N=3000, i_v3_0", "i_v3_0", [(X, ids0, q, M) for X, ids0, q, M in synth3]))
    for it in range(3):
        out.append((f"fuzz leg {it} (ragged random clouds), i_v4_0", "i_v4_0", fp.leg_round(it)["structs"]))
    return out


def main():
    limits = [8.0, 16.0, 24.0, 32.0, 40.0, 48.0, 64.0, 96.0, 0.0]
    sets = datasets()
    print("# structures repeated on the exact fp32 kernels by precision 'auto' (of the set's size) per state limit (0 = trigger off)")
    print(f"{'data set':58s} " + " ".join(f"{l:6.0f}" for l in limits))
    models = {}
    for name, tag, structs in sets:
        row = []
        for lim in limits:
            m = models.get(tag)
            if m is None:
                m = Model(CONFIGS[tag]); m.load_state_dict(weights(tag)); models[tag] = m
            m.set_auto_state_limit(lim)
            before = m.status()["n_fp32_rerun"]
            m.forward_batch(structs, independent=True)
            row.append(m.status()["n_fp32_rerun"] - before)
        print(f"{name:58s} " + " ".join(f"{r:6d}" for r in row) + f"   of {len(structs)}", flush=True)
    # the pinned inputs: deviation from the reference's fp64 logits per limit
    for pre in ("a_", "b_"):
        p = fp.pinned(pre)
        row = []
        for lim in limits:
            m = models["i_v4_0"]; m.set_auto_state_limit(lim)
            before = m.status()["n_fp32_rerun"]
            z = m.forward_segments(p["X"], p["ids"], p["q0"], p["roa"], p["R"])
            row.append((float(np.abs(z - p["z64"]).max()), float(np.abs(z - p["z32"]).max()), m.status()["n_fp32_rerun"] - before))
        print(f"pinned {pre} |hip-z64| / |hip-z32_t8| / repeated:  " + "  ".join(f"{a:.1e}/{b:.1e}/{c}" for a, b, c in row)
              + f"   (reference fp32 vs fp64 {np.abs(p['z32'] - p['z64']).max():.2e})", flush=True)


if __name__ == "__main__":
    main()
