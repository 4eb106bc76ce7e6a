"""developer script: one layer through the rendezvous mode (1) and the 32-edge-tile kernel (3), error structure of the difference"""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import golden, weights
from pesto_amd import Model
from pesto_amd.config import CONFIGS

g = golden("ops_i_v4_0_crop200")
for layer in (4, 8, 12):
    out = {}
    for mode in (1, 3):
        m = Model(CONFIGS["i_v4_0"], precision="f16_split").debug_edge_mode(mode)
        m.load_state_dict(weights("i_v4_0"))
        m.stage_unpack(g["X"], g["ids_topk"].astype(np.int32))
        out[mode] = m.stage_layer(layer, g[f"L{layer}_q_in"], g[f"L{layer}_p_in"])
    qr, pr = g[f"L{layer}_q_out"], g[f"L{layer}_p_out"]
    dq, dp = out[3][0] - out[1][0], out[3][1] - out[1][1]
    print(f"layer {layer} nn {CONFIGS['i_v4_0']['sum'][layer]['nn']}: mode1 vs golden q {np.abs(out[1][0]-qr).max():.2e} p {np.abs(out[1][1]-pr).max():.2e} | mode3-mode1 q {np.abs(dq).max():.3e} p {np.abs(dp).max():.3e} finite {np.isfinite(out[3][0]).all()}")
    ea = np.abs(dq).max(1)
    print("   q err by atom (first 12):", np.array2string(ea[:12], precision=3), " even/odd mean", ea[0::2].mean(), ea[1::2].mean())
    print("   q err by feature (first 8):", np.array2string(np.abs(dq).max(0)[:8], precision=3), " p err by xyz", np.abs(dp).max((0, 2)))
    print("   rel: |dq|/|q| ", np.abs(dq).mean() / np.abs(out[1][0]).mean(), " |dp|/|p|", np.abs(dp).mean() / np.abs(out[1][1]).mean())
