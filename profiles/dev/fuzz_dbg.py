import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "profiles"))
import importlib.util
src = open(os.path.join(ROOT, "profiles", "fuzz_vs_oracle.py")).read().split("worst = 0.0")[0]
exec(src)
sizes = [500, 3]
structs = [structure(n, 1000 * 0 + j) for j, n in enumerate(sizes)]
Xc, idc, qc, Mc = collate_batch_features([list(s) for s in structs])
roa_c, R_c = mask_to_segments(Mc)
z_ref = o.forward_segments(Xc, idc, qc, roa_c, R_c)
for prec in ("auto", "fp32"):
    m.set_precision(prec)
    z_col = np.concatenate(m.forward_batch(structs, independent=False), 0)
    err = np.abs(z_col - z_ref)
    r = np.unravel_index(err.argmax(), err.shape)
    print(prec, "max err", err.max(), "at residue", r, "of", z_ref.shape, "z_ref there", z_ref[r], "hip", z_col[r], "|z|max", np.abs(z_ref).max(),
          "err on the 500-atom part", err[:structs[0][3].shape[1]].max(), "on the 3-atom part", err[structs[0][3].shape[1]:].max())
