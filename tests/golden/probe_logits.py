#!/usr/bin/env python3
"""Range of the attention logits the REFERENCE feeds its softmaxes (build container only; imports /root/reference).

The layer kernel evaluates softmax as exp2(t) / sum exp2(t) WITHOUT subtracting the row maximum (src/model_operations.py:139-140 goes
through torch's max-subtracted softmax; the two are the same function). That is only safe while every logit stays far inside the
fp32 exponent range; the kernel guards the assumption (|t| > LOGIT_GUARD flags the structure for the exact kernels). This probe
records what the trained checkpoints actually produce: max |logit| and the smallest row maximum over every softmax call of a forward.

  python tests/golden/probe_logits.py  ->  profiles/r05_logit_range.txt
"""
import os
import sys

import numpy as np
import torch as pt

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from pesto_amd.topology import collate_batch_features, extract_topology  # noqa: E402
from pesto_amd.weights import stack_layers  # noqa: E402

stats = {}


def patched_softmax(x, dim=None, **kw):
    s = stats.setdefault("cur", {"max_abs": 0.0, "min_rowmax": 1e30, "max_rowmax": -1e30, "calls": 0})
    rm = x.max(dim=dim).values
    s["max_abs"] = max(s["max_abs"], float(x.abs().max()))
    s["min_rowmax"] = min(s["min_rowmax"], float(rm.min()))
    s["max_rowmax"] = max(s["max_rowmax"], float(rm.max()))
    s["calls"] += 1
    return _orig(x, dim=dim, **kw)


_orig = pt.nn.functional.softmax


def run(tag, model, X, ids, q, M):
    stats.pop("cur", None)
    pt.nn.functional.softmax = patched_softmax
    try:
        with pt.no_grad():
            model(pt.from_numpy(X), pt.from_numpy(ids.astype(np.int64)), pt.from_numpy(q), pt.from_numpy(M))
    finally:
        pt.nn.functional.softmax = _orig
    s = stats["cur"]
    line = f"{tag:44s} softmax calls {s['calls']:3d}  max|logit| {s['max_abs']:9.3f}  row maxima in [{s['min_rowmax']:9.3f}, {s['max_rowmax']:8.3f}]"
    print(line, flush=True)
    return line


def chain_inputs(d, i, n0_cols):
    a0, a1 = int(d["atom_offsets"][i]), int(d["atom_offsets"][i + 1])
    X = d["X"][a0:a1]
    roa = d["res_of_atom"][a0:a1].astype(np.int64)
    roa = roa - roa.min()
    R = int(roa.max()) + 1
    if n0_cols == 30:
        q = np.zeros((a1 - a0, 30), np.float32); q[np.arange(a1 - a0), d["q_idx"][a0:a1, 0]] = 1
    else:
        q = np.zeros((a1 - a0, 123), np.float32)
        for c, off in enumerate((0, 30, 59)):
            q[np.arange(a1 - a0), off + d["q_idx3"][a0:a1, c]] = 1
    M = np.zeros((a1 - a0, R), np.float32); M[np.arange(a1 - a0), roa] = 1
    ids = extract_topology(X, 64)
    Xc, idc, qc, Mc = collate_batch_features([(X, ids, q, M)])
    return Xc, idc, qc, Mc


def main():
    out = []
    d = np.load(os.path.join(HERE, "cfg4_all53.npz"))
    picks = [0, 13, 26, 41, 52]
    for run_name, n0 in (("i_v4_0_2021-09-07_11-20", 30), ("i_v3_0_2021-05-27_14-27", 123), ("i_v3_1_2021-05-28_12-40", 123)):
        try:
            cfg, model = mg.load_run(run_name)
        except Exception as ex:      # (a run that is absent upstream)
            out.append(f"{run_name}: not loadable here ({type(ex).__name__})")
            continue
        for i in picks:
            out.append(run(f"{run_name[:6]} {d['names'][i].decode()}", model, *chain_inputs(d, i, n0)))
    # the i_v4_1 architecture with the stacked weights every i_v4_1 golden uses
    cfg0, model0 = mg.load_run("i_v4_0_2021-09-07_11-20")
    sd0 = {k: v.numpy() for k, v in model0.state_dict().items()}
    cfg1, Model1, _ = mg.import_reference("i_v4_1_2021-09-07_11-21")
    m1 = Model1(cfg1).eval()
    sd1 = stack_layers(sd0, cfg1, 0.5)
    m1.load_state_dict({k: pt.from_numpy(np.asarray(v)) for k, v in sd1.items()}, strict=False)
    for i in picks:
        out.append(run(f"i_v4_1 stacked {d['names'][i].decode()}", m1, *chain_inputs(d, i, 30)))
    g = np.load(os.path.join(HERE, "fuzz_pins.npz"))
    print(g.files)
    hdr = ("# attention logits of the REFERENCE's softmaxes (tests/golden/probe_logits.py: the reference imported in the build container,\n"
           "# torch.nn.functional.softmax patched to record its inputs). 'row maxima' = smallest and largest per-row maximum over the attention\n"
           "# softmaxes of a forward (max|logit| ~ 1e6 is the pool layer's -1e6 mask, src/model_operations.py:199). The kernel's exp2 / sum without max\n"
           "# subtraction needs them inside fp32's exponent range and guards that per centre (DESIGN 4.1h).\n")
    open(os.path.join(ROOT, "profiles", "r05_logit_range.txt"), "w").write(hdr + "\n".join(out) + "\n")


if __name__ == "__main__":
    main()
