import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from conftest import cfg4_all53, example_complexes, weights
from pesto_amd import Model
from pesto_amd.config import CONFIGS
from pesto_amd.topology import _norm_xyz
m = Model(CONFIGS["i_v4_0"]); m.load_state_dict(weights("i_v4_0"))
tot = 0
for ch in (example_complexes() if 'examples' in sys.argv else cfg4_all53()):
    X = ch["X"]
    g = np.asarray(m.knn_collate(X, [X.shape[0]])) - 1
    h = ch["ids0_host"].astype(np.int64)
    bad = np.argwhere(g != h)
    if len(bad):
        print(ch["name"], "differing entries", len(bad))
        print("   rows", sorted(set(bad[:, 0]))[:12], "slots", sorted(set(bad[:, 1]))[:12])
        for r, c in bad[:6]:
            a, b = g[r, c], h[r, c]
            da, db = _norm_xyz((X[[a, b]] - X[r])[None])[0]
            x64 = X.astype(np.float64)
            print("  row", r, "slot", c, "gpu", a, repr(da), "host", b, repr(db), "f64:", np.linalg.norm(x64[a]-x64[r]), np.linalg.norm(x64[b]-x64[r]),
                  "same set", sorted(g[r]) == sorted(h[r]))
        tot += len(bad)
print("total differing", tot)
