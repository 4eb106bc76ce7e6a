import sys, os, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_fuzz_pins as fp
from conftest import golden
g = golden("fuzz_pins")
for pre in ("a_", "b_"):
    p = fp.pinned(pre)
    for prec in ("auto", "fp32"):
        m = fp._model(prec)
        z = m.forward_segments(p["X"], p["ids"], p["q0"], p["roa"], p["R"])
        err = np.abs(z - p["z64"]).max(1)
        e32 = np.abs(p["z32"] - p["z64"]).max(1)
        order = np.argsort(-err)[:5]
        print(pre, prec, "max", err.max(), "worst residues", [(int(r), float(err[r]), float(e32[r]), float(np.abs(p["z64"][r]).max())) for r in order], "n>1e-4:", int((err > 1e-4).sum()), "of", err.size, "median", float(np.median(err)))
    if pre == "a_":
        sizes = g["a_sizes"]; print("sizes", sizes.tolist())
