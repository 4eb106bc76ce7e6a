"""Where a layer launch spends its time OUTSIDE the steady item loop: wall-clock stamps (s_memrealtime, 100 MHz) of every workgroup of every
layer launch of ONE forward of the bench workload, from a developer build:
    PESTO_LIB_TAG=tl PESTO_EXTRA_CXXFLAGS=-DPESTO_DEV_TIMELINE python -m pesto_amd.csrc.build
    PESTO_LIB=$PWD/pesto_amd/csrc/libpesto_hip_tl.so python profiles/dev/timeline.py
stamps per workgroup: 0 kernel entry, 1 constants staged (first barrier), 2 item wave 0 leaves, 4 item wave 7 leaves, 3 node wave role 0 leaves,
5 node wave role 1 leaves (node-wave kernels only)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from pesto_amd import Model, CONFIGS, _lib

dev = torch.device("cuda:0")
cfg = CONFIGS["i_v4_1"]
sd, _ = bench.load_weights(cfg)
m = Model(cfg, validate=False, precision="f16_split").to(dev)
m.load_state_dict(sd)
batch = int(os.environ.get("TL_BATCH", "8"))
X, ids, q, roa, R = bench.make_batch(3000, batch, 1, 30)
a = [torch.from_numpy(v).to(dev) for v in (X, ids, q, roa)] + [R]
lib = _lib.load()
fn = lib.pesto_dev_timeline
fn.argtypes = [ctypes.c_void_p]; fn.restype = ctypes.c_int
buf = np.zeros((64, 256, 16), dtype=np.uint64)
for _ in range(30): m.forward_segments(*a)
torch.cuda.synchronize()
fn(buf.ctypes.data)
m.forward_segments(*a)
torch.cuda.synchronize()
n = fn(buf.ctypes.data)
nns = [l["nn"] for l in cfg["sum"]]
print(f"launches seen {n}; times in us (10 ns stamps), mean over the workgroups of a launch unless said otherwise")
rdv = {}; chain = {}; acc = {}; by_xcd = {}; in_xcd = {}; by_jb = {}
for l in range(min(n, 64)):
    t = buf[l].astype(np.int64)
    on = t[:, 0] > 0
    if not on.any():
        continue
    t = t[on]
    t0 = t[:, 0].min()
    rendezvous = (t[:, 15] > 0).all()      # (slot 15 = wave 11's arrival at a rendezvous: node-wave kernels never write it)
    nodew = not rendezvous and (t[:, 3] > 0).all()
    item_end = np.maximum(t[:, 2], t[:, 4])
    end = np.maximum(item_end, np.maximum(t[:, 3], t[:, 5])) if nodew else item_end
    row = {"wgs": int(on.sum()), "span": (end.max() - t0) / 100.0, "entry_spread": (t[:, 0].max() - t0) / 100.0, "prologue": (t[:, 1] - t[:, 0]).mean() / 100.0,
           "items": (item_end - t[:, 1]).mean() / 100.0, "node_tail": ((end - item_end).mean() / 100.0) if nodew else 0.0,
           "first_wg_end": (end.min() - t0) / 100.0, "last_wg_end": (end.max() - t0) / 100.0, "mean_wg_end": (end.mean() - t0) / 100.0}
    nn = nns[l] if l < len(nns) else 0
    acc.setdefault(nn, []).append(row)
    if rendezvous:      # rendezvous mode: arrivals of the twelve waves at the LAST rendezvous, release seen by wave 0
        arr = t[:, 4:16]
        last = arr.max(axis=1)
        rdv.setdefault(nn, []).append([((last[:, None] - arr).mean() / 100.0), ((last - arr.min(axis=1)).mean() / 100.0), ((t[:, 3] - last).mean() / 100.0),
                                       ((last[:, None] - arr).reshape(len(arr), 3, 4).mean(axis=(0, 1)) / 100.0).tolist()])
    if nodew:      # the LAST tile of node waves role 0 / role 1, relative to the moment the last item wave left
        ie = item_end
        chain.setdefault(nn, []).append([((t[:, k] - ie).mean() / 100.0) for k in (6, 7, 8, 9, 10, 3, 11, 12, 13, 14, 5)])
    if on.all():
        e = (end - t0) / 100.0
        by_xcd.setdefault(nn, []).append([e[x::8].mean() for x in range(8)])
        in_xcd.setdefault(nn, []).append(np.mean([e[x::8].max() - e[x::8].mean() for x in range(8)]))
        by_jb.setdefault(nn, []).append([e[8 * j:8 * j + 8].mean() for j in range(32)])
for nn in sorted(acc):
    rows = acc[nn]
    print(f"nn = {nn:2d} ({len(rows)} launches, {rows[0]['wgs']} workgroups): " + "  ".join(f"{k} {np.mean([r[k] for r in rows]):6.1f}" for k in rows[0] if k != "wgs"))
for nn in sorted(by_xcd):
    print(f"nn = {nn:2d} mean end by XCD:", " ".join(f"{v:6.1f}" for v in np.mean(by_xcd[nn], axis=0)), f"| max - mean inside an XCD {np.mean(in_xcd[nn]):5.1f}")
    print(f"        mean end by workgroup index within the XCD:", " ".join(f"{v:5.0f}" for v in np.mean(by_jb[nn], axis=0)))
for nn in sorted(chain):
    v = np.mean(chain[nn], axis=0)
    print(f"nn = {nn:2d} last tile, us after the last item wave left: role 0 loop top {v[0]:5.1f} rows {v[1]:5.1f} state done {v[2]:5.1f} inputs {v[3]:5.1f} [U|A] done {v[4]:5.1f} end {v[5]:5.1f} | "
          f"role 1 loop top {v[6]:5.1f} rows {v[7]:5.1f} state + G done {v[8]:5.1f} inputs {v[9]:5.1f} end {v[10]:5.1f}")
for nn in sorted(rdv):
    v = rdv[nn]
    print(f"nn = {nn:2d} last rendezvous: a wave waits {np.mean([r[0] for r in v]):5.2f} us on average for the last of its workgroup (first to last arrival {np.mean([r[1] for r in v]):5.2f} us; "
          f"release seen {np.mean([r[2] for r in v]):4.2f} us after the last arrival); by SIMD (wave % 4): " + " ".join(f"{x:5.2f}" for x in np.mean([r[3] for r in v], axis=0)))
