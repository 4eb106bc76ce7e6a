"""developer script (needs libpesto_hip_dbg32.so, PESTO_LIB): intermediates of one centre in the 32-edge-tile kernel against a numpy
restatement of StateUpdate.forward (src/model_operations.py:87-154) for that centre."""
import ctypes
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import golden, weights
from pesto_amd import Model, _lib
from pesto_amd.config import CONFIGS
from oracle import oracle

layer = int(sys.argv[1]) if len(sys.argv) > 1 else 4
centre = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cfg = CONFIGS["i_v4_0"]
nn = cfg["sum"][layer]["nn"]
g = golden("ops_i_v4_0_crop200")
sd = weights("i_v4_0")
q_in, p_in = g[f"L{layer}_q_in"], g[f"L{layer}_p_in"]          # [N1,32], [N1,3,32]
ids_s, D, R = oracle.OracleModel.unpack(g["X"], g["ids_topk"].astype(np.int32))


def elu(x):
    return np.where(x > 0, x, np.expm1(np.minimum(x, 0)))


def mlp(x, pre):
    for i in (0, 2, 4):
        x = x @ sd[f"{pre}.{i}.weight"].astype(np.float64).T + sd[f"{pre}.{i}.bias"].astype(np.float64)
        if i < 4:
            x = elu(x)
    return x


i = centre
nb = ids_s[i, :nn]
q, p = q_in[i].astype(np.float64), p_in[i].astype(np.float64)
qn, pn = q_in[nb].astype(np.float64), p_in[nb].astype(np.float64)          # [n,32], [n,3,32]
d, r = D[i, :nn].astype(np.float64), R[i, :nn].astype(np.float64)
Xn = np.concatenate([q, np.linalg.norm(p, axis=0)])
Xe = np.concatenate([d[:, None], np.repeat(Xn[None], nn, 0), qn, np.linalg.norm(pn, axis=1), (p[None] * r[:, :, None]).sum(1), (pn * r[:, :, None]).sum(1)], 1)
pre = f"sum.{layer}.su"
Q = mlp(Xn, pre + ".nqm").reshape(2, 2, 3)
Kq = mlp(Xe, pre + ".eqkm")                                                # [n,3]
Kp = mlp(Xe, pre + ".epkm")                                                # [n,9] -> parts 1..3 = columns 0:3, 3:6, 6:9
V = mlp(Xe, pre + ".evm").reshape(nn, 2, 32)
sdk = np.sqrt(3.0)
logit = np.zeros((2, 4, nn))
for h in range(2):
    logit[h, 0] = Kq @ Q[0, h] / sdk
    for part in range(3):
        logit[h, 1 + part] = Kp[:, 3 * part:3 * part + 3] @ Q[1, h] / sdk
ms = logit[:, 0].max(1)
mv = logit[:, 1:].reshape(2, -1).max(1)
es = np.exp(logit[:, 0] - ms[:, None])
ev = np.exp(logit[:, 1:] - mv[:, None, None])
Ss, Sv = es.sum(1), ev.reshape(2, -1).sum(1)
Zq = np.einsum("he,es->hs", es / Ss[:, None], V[:, 0])
Zp = np.zeros((3, 2, 32))
for h in range(2):
    w = ev[h] / Sv[h]
    for c in range(3):
        Zp[c, h] = (w[0] * r[:, c]) @ V[:, 1] + w[1].sum() * p[c] + w[2] @ pn[:, c]

# ---- the kernel
m = Model(cfg, precision="f16_split").debug_edge_mode(3)
m.load_state_dict(sd)
lib = _lib.load()
lib.pesto_debug_dump32.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.pesto_debug_dump32(None, centre)
m.stage_unpack(g["X"], g["ids_topk"].astype(np.int32))
try:
    m.stage_layer(layer, q_in, p_in)
except Exception as e:
    print("stage_layer:", e)
buf = np.zeros(4096, np.float32)
lib.pesto_debug_dump32(buf.ctypes.data_as(ctypes.c_void_p), -1)
A = 2 if nn <= 32 else 1
c0 = centre // A * A
cs = centre - c0 if nn == 16 else 0
n_tiles = 2 if nn == 64 else 1
np.set_printoptions(precision=4, linewidth=200, suppress=True)
for t in range(n_tiles):
    tt = t if nn != 32 else centre - c0           # nn = 32: the centre's tile within the item
    la = buf[(tt * 8 + 0) * 64:(tt * 8 + 2) * 64].reshape(2, 64)
    lb = buf[(tt * 8 + 2) * 64:(tt * 8 + 4) * 64].reshape(2, 64)
    cols = np.arange(16) + 16 * cs if nn == 16 else np.arange(32)
    e_of = (np.arange(len(cols)) + 32 * t) if nn != 16 else np.arange(16)
    for h in range(2):
        got = np.stack([la[h, cols], la[h, 32 + cols], lb[h, cols], lb[h, 32 + cols]])
        ref = logit[h][:, e_of]
        print(f"tile {t} head {h}: max |logit - ref| per part", np.abs(got - ref).max(1), " ref range", ref.min(), ref.max())
st = buf[1024:1280].reshape(64, 4)
print("stat lane0 (1/Ss, 1/Sv, W2/Sv, cs):", st[0], " lane32:", st[32], "   ref h0:", 1 / Ss[0], 1 / Sv[0], ev[0, 1].sum() / Sv[0], " h1:", 1 / Ss[1], 1 / Sv[1], ev[1, 1].sum() / Sv[1])
blk = buf[1280 + cs * 512:1280 + cs * 512 + 512]
tq = blk[:128].reshape(2, 64)
print("tq (unnormalised Zq sums) lane 0..3 h0:", tq[0, :4], " ref:", (es[0] @ V[:, 0])[:4], "   h1:", tq[1, :4], (es[1] @ V[:, 0])[:4])
tp0 = blk[128:320].reshape(3, 64)
print("tp[h=0][c] lane 0..3:", tp0[:, :4].ravel(), " ref part-1 sums:", np.array([(ev[0, 0] * r[:, c]) @ V[:, 1] for c in range(3)])[:, :4].ravel())
pown = blk[320:512].reshape(3, 64)
print("p_own lane 0..3:", pown[:, :4].ravel(), " ref:", p[:, :4].ravel())
slot = cs if nn == 16 else (centre - c0 if nn == 32 else 0)
zb = buf[2304 + (cs if nn == 16 else 0) * 256:2304 + (cs if nn == 16 else 0) * 256 + 256]
print("Zq err", np.abs(zb[:64].reshape(2, 32) - Zq).max(), " Zp err", np.abs(zb[64:].reshape(3, 2, 32) - Zp).max(), " |Zq|max", np.abs(Zq).max(), "|Zp|max", np.abs(Zp).max())
print("Zp got[0,0,:4]", zb[64:68], " ref", Zp[0, 0, :4])
