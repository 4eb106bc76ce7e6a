#!/usr/bin/env python3
"""Generate the golden input/output vectors under tests/golden/ by IMPORTING the
reference PyTorch CPU path from /root/reference (build container only).

The reference python never leaves this container: what is committed is DATA
(inputs, weights-as-arrays, expected outputs) plus this script.  Nothing under
tests/, bench.py or __graft_entry__.py reads /root/reference at run time.

Reference entry points exercised (file:line relative to /root/reference):
  model/model.py:32-52              Model.forward  (whole-forward goldens)
  src/model_operations.py:6-22      unpack_state_features  (op golden)
  src/model_operations.py:225-242   StateUpdateLayer.forward (op goldens, nn=8/16/32/64)
  src/model_operations.py:197-213   StatePoolLayer.forward  (op golden)
  src/data_encoding.py:61-102       encode_structure / encode_features / extract_topology
  src/dataset.py:91-112             collate_batch_features

Usage:  python tests/golden/make_golden.py        (takes a few minutes on 8 cores)
"""
import os
import sys
import types
import warnings

import numpy as np
import torch as pt

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
warnings.filterwarnings("ignore")
pt.manual_seed(0)


def import_reference(run):
    """Mirror apply_model.ipynb:67-73: put the saved run first on sys.path, then the repo root."""
    for m in ("config", "model"):
        sys.modules.pop(m, None)
    save_path = os.path.join(REF, "model", "save", run)
    sys.path = [p for p in sys.path if "/model/save/" not in p]
    sys.path.insert(0, save_path)
    if REF not in sys.path:
        sys.path.insert(1, REF)
    if "gemmi" not in sys.modules:  # src/structure_io.py:1 imports gemmi (absent here); never called
        g = types.ModuleType("gemmi")
        g.cif = types.ModuleType("gemmi.cif")
        sys.modules["gemmi"] = g
        sys.modules["gemmi.cif"] = g.cif
    from config import config_model
    from model import Model
    return config_model, Model, save_path


def load_run(run, ckpt="model_ckpt.pt"):
    config_model, Model, save_path = import_reference(run)
    model = Model(config_model)
    sd = pt.load(os.path.join(save_path, ckpt), map_location="cpu")
    print(run, model.load_state_dict(sd))
    return config_model, model.eval()


def parse_pdb(path):
    """Fixed-column reader for the already-cleaned single-model heavy-atom files under
    pdbs_test/ and examples/ (columns as written by src/structure_io.py:118)."""
    name, resname, resid, xyz, element, chain = [], [], [], [], [], []
    for line in open(path):
        if not (line.startswith("ATOM") or line.startswith("HETATM")):
            continue
        name.append(line[12:16].strip())
        resname.append(line[17:20].strip())
        chain.append(line[21])
        resid.append(int(line[22:26]))
        xyz.append([float(line[30:38]), float(line[38:46]), float(line[46:54])])
        element.append(line[76:78].strip().capitalize())
    structure = {
        "xyz": np.array(xyz, dtype=np.float32),
        "name": np.array(name),
        "element": np.array(element),
        "resname": np.array(resname),
        "resid": np.array(resid, dtype=np.int32),
        "chain_name": np.array(chain),
    }
    # renumber residues contiguously the way clean_structure does (src/structure.py:33-50)
    d_chain = np.concatenate([[0], (structure["chain_name"][1:] != structure["chain_name"][:-1]).astype(int)])
    d_res = np.abs(np.sign(np.concatenate([[0], np.diff(structure["resid"])])))
    structure["resid"] = (np.cumsum(np.sign(d_chain + d_res)) + 1).astype(np.int64)
    return structure


def encode(structure, all_features):
    """apply_model.ipynb:141-152 (reference functions, imported)."""
    from src.data_encoding import encode_structure, encode_features, extract_topology
    X, M = encode_structure(structure)
    qs = encode_features(structure)
    q = pt.cat(qs, dim=1) if all_features else qs[0]
    ids_topk = extract_topology(X, 64)[0]
    return X, ids_topk, q, M


def collate(items):
    from src.dataset import collate_batch_features
    return collate_batch_features(items)


def onehot_to_idx(q, all_features):
    q = q.numpy()
    if all_features:
        return np.stack([q[:, :30].argmax(1), q[:, 30:59].argmax(1), q[:, 59:].argmax(1)], 1).astype(np.int16)
    return q.argmax(1).astype(np.int16)[:, None]


def res_of_atom(M):
    M = M.numpy()
    assert np.all(M.sum(1) == 1)
    return M.argmax(1).astype(np.int32)


def sd_arrays(model):
    return {k: v.detach().numpy() for k, v in model.state_dict().items()}


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"  wrote {name}.npz  {os.path.getsize(path) / 1024:.0f} KiB")


def synth_inputs(n, seed, n0=30):
    """Synthetic structure from the package's seeded generator (so bench/tests can rebuild the same
    cloud from the seed); topology from the REFERENCE's extract_topology."""
    from src.data_encoding import extract_topology
    from pesto_amd.topology import synthetic_structure
    X, _, q, M = synthetic_structure(n, seed, n0=n0)
    X = pt.from_numpy(X)
    return X, extract_topology(X, 64)[0], pt.from_numpy(q), pt.from_numpy(M)


def run_forward(model, X, ids, q, M):
    with pt.no_grad():
        return model(X, ids, q, M.float()).numpy()


def main():
    # ------------------------------------------------------------------ weights (real, as arrays)
    models = {}
    for tag, run in (("i_v4_0", "i_v4_0_2021-09-07_11-20"), ("i_v3_0", "i_v3_0_2021-05-27_14-27"),
                     ("i_v3_1", "i_v3_1_2021-05-28_12-40")):
        cfg, model = load_run(run)
        if tag == "i_v3_1":
            # The trained i_v3_1 is numerically chaotic on these inputs: the REFERENCE's own fp32 and fp64 runs
            # differ by 10.9 on z (states reach 4e5), so it cannot anchor a 1e-4 parity test. Its ARCHITECTURE
            # (single-Linear em and dm, model/save/i_v3_1_*/model.py:10-22) is pinned with a hybrid instead:
            # i_v3_1's own em/dm weights + i_v3_0's sum/spl weights (same shapes; reference fp32 vs fp64: 7e-7).
            sd30 = models["i_v3_0"][1].state_dict()
            sd = {k: (sd30[k] if k.startswith(("sum.", "spl.")) else v) for k, v in model.state_dict().items()}
            print("i_v3_1 hybrid", model.load_state_dict(sd))
            save("weights_i_v3_1_emdm", **{k: v for k, v in sd_arrays(model).items() if k.startswith(("em.", "dm."))})
            models[tag] = (cfg, model)
            continue
        models[tag] = (cfg, model)
        save("weights_" + tag, **sd_arrays(model))

    pdb_2ayo = parse_pdb(os.path.join(REF, "pdbs_test", "AY_2AYO_1_A:0.pdb"))
    pdb_2cua = parse_pdb(os.path.join(REF, "examples", "issue_19_04_2023", "2CUA_A.pdb"))
    print("2AYO atoms", pdb_2ayo["xyz"].shape[0], "2CUA atoms", pdb_2cua["xyz"].shape[0])

    # ------------------------------------------------------------------ whole-forward, real weights, real geometry
    for tag in ("i_v4_0", "i_v3_0", "i_v3_1"):
        cfg, model = models[tag]
        import_reference({"i_v4_0": "i_v4_0_2021-09-07_11-20", "i_v3_0": "i_v3_0_2021-05-27_14-27",
                          "i_v3_1": "i_v3_1_2021-05-28_12-40"}[tag])
        allf = cfg["em"]["N0"] == 123
        for pname, st in (("2AYO", pdb_2ayo), ("2CUA", pdb_2cua)):
            if tag != "i_v4_0" and pname == "2AYO":
                continue  # keep the CPU suite short: 2AYO only with i_v4_0
            X, ids, q, M = encode(st, allf)
            Xc, idsc, qc, Mc = collate([[X, ids, q, M]])
            z = run_forward(model, Xc, idsc, qc, Mc)
            save(f"fwd_{tag}{'h' if tag == 'i_v3_1' else ''}_{pname}", X=Xc.numpy(), ids_topk=idsc.numpy().astype(np.int32),
                 q_idx=onehot_to_idx(qc, allf), res_of_atom=res_of_atom(Mc), z=z)

    cfg40, m40 = models["i_v4_0"]
    import_reference("i_v4_0_2021-09-07_11-20")

    # ------------------------------------------------------------------ per-op goldens on a 200-atom crop of 2CUA
    crop = {k: v[:200] for k, v in pdb_2cua.items()}
    X, ids, q, M = encode(crop, False)
    Xc, idsc, qc, Mc = collate([[X, ids, q, M]])
    from src.model_operations import unpack_state_features
    rec = {}
    with pt.no_grad():
        q1 = m40.em(qc)
        rec["em_in_idx"] = onehot_to_idx(qc, False)
        rec["em_out"] = q1.numpy()
        qs, ids_s, D, R = unpack_state_features(Xc, idsc, q1)
        rec.update(X=Xc.numpy(), ids_topk=idsc.numpy().astype(np.int32), D_nn=D.numpy(), R_nn=R.numpy())
        p = pt.zeros((qs.shape[0], 3, qs.shape[1]))
        state = (qs, p, ids_s, D, R)
        for li, layer in enumerate(m40.sum):
            if li in (0, 3, 4, 8, 12, 15):
                rec[f"L{li}_q_in"] = state[0].clone().numpy()
                rec[f"L{li}_p_in"] = state[1].clone().numpy()
            state = layer(state)
            state = tuple(t.detach() for t in state)
            if li in (0, 3, 4, 8, 12, 15):
                rec[f"L{li}_q_out"] = state[0].clone().numpy()
                rec[f"L{li}_p_out"] = state[1].clone().numpy()
        qa, pa = state[0], state[1]
        qr, pr = m40.spl(qa[1:], pa[1:], Mc.float())
        rec["pool_qr"] = qr.numpy()
        rec["pool_pr"] = pr.numpy()
        zr = pt.cat([qr, pt.norm(pr, dim=1)], dim=1)
        rec["z"] = m40.dm(zr).numpy()
        rec["res_of_atom"] = res_of_atom(Mc)
    save("ops_i_v4_0_crop200", **rec)

    # ------------------------------------------------------------------ edge cases (i_v4_0 real weights)
    # (a) N=40 < 64: zero-padded ids -> X[-1] wrap-around (model_operations.py:8)
    small = {k: v[:40] for k, v in pdb_2cua.items()}
    a = encode(small, False)
    Xc, idsc, qc, Mc = collate([list(a)])
    save("edge_n40", X=Xc.numpy(), ids_topk=idsc.numpy().astype(np.int32), q_idx=onehot_to_idx(qc, False),
         res_of_atom=res_of_atom(Mc), z=run_forward(m40, Xc, idsc, qc, Mc))
    # (b) two-structure batch (N=300 + N=40), block-diagonal M, offset ids (dataset.py:102-110)
    b0 = encode({k: v[:300] for k, v in pdb_2cua.items()}, False)
    b1 = encode({k: v[300:340] for k, v in pdb_2cua.items()}, False)
    Xc, idsc, qc, Mc = collate([list(b0), list(b1)])
    save("edge_batch2", X=Xc.numpy(), ids_topk=idsc.numpy().astype(np.int32), q_idx=onehot_to_idx(qc, False),
         res_of_atom=res_of_atom(Mc), z=run_forward(m40, Xc, idsc, qc, Mc),
         sizes=np.array([[300, int(b0[3].shape[1])], [40, int(b1[3].shape[1])]], dtype=np.int32))
    # (c) coincident atoms: duplicate 3 atoms exactly -> D<1e-2 -> global-max fix-up (model_operations.py:12)
    co = {k: v[:150].copy() for k, v in pdb_2cua.items()}
    co["xyz"][10] = co["xyz"][11]
    co["xyz"][77] = co["xyz"][20] + np.float32(1e-3)
    c = encode(co, False)
    Xc, idsc, qc, Mc = collate([list(c)])
    # force a coincident atom INTO the neighbour list (extract_topology pushes them to the far end)
    idsc = idsc.clone()
    idsc[10, 5] = 12   # atom 11 (1-based 12) has the same coordinates as atom 10
    idsc[20, 63] = 78  # atom 77 is 1e-3*sqrt(3) A from atom 20
    save("edge_coincident", X=Xc.numpy(), ids_topk=idsc.numpy().astype(np.int32), q_idx=onehot_to_idx(qc, False),
         res_of_atom=res_of_atom(Mc), z=run_forward(m40, Xc, idsc, qc, Mc))
    # (d) residue with ONE atom + a big residue
    sr = {k: v[:120].copy() for k, v in pdb_2cua.items()}
    sr["resid"] = sr["resid"].copy()
    sr["resid"][0] = 0               # own residue
    sr["resid"][60:] = 1000          # one 60-atom residue
    d = encode(sr, False)
    Xc, idsc, qc, Mc = collate([list(d)])
    save("edge_single_atom_residue", X=Xc.numpy(), ids_topk=idsc.numpy().astype(np.int32),
         q_idx=onehot_to_idx(qc, False), res_of_atom=res_of_atom(Mc), z=run_forward(m40, Xc, idsc, qc, Mc))

    # ------------------------------------------------------------------ i_v4_1 ARCHITECTURE (weights blob missing upstream)
    cfg41, Model41, _ = import_reference("i_v4_1_2021-09-07_11-21")
    assert len(cfg41["sum"]) == 32
    m41 = Model41(cfg41).eval()
    # stacked real weights (pesto_amd.weights.stack_layers): every i_v4_0 layer duplicated with its
    # residual branch halved -> 32 layers with realistic activation magnitudes (|z| <~ 25)
    from pesto_amd.weights import stack_layers
    sd40 = {k: v.numpy() for k, v in m40.state_dict().items()}
    sd41 = stack_layers(sd40, cfg41, residual_scale=0.5)
    print("i_v4_1 stacked", m41.load_state_dict({k: pt.from_numpy(np.array(v)) for k, v in sd41.items()}))
    for n, seed in ((512, 3), (3000, 1)):
        X, ids, q, M = synth_inputs(n, seed)
        Xc, idsc, qc, Mc = collate([[X, ids, q, M]])
        z = run_forward(m41, Xc, idsc, qc, Mc)
        print(f"  i_v4_1 stacked N={n}: |z|max={np.abs(z).max():.3f} finite={np.isfinite(z).all()}")
        save(f"fwd_i_v4_1_stacked_synth{n}", z=z, seed=np.int64(seed), ids_topk=idsc.numpy().astype(np.int16 if n < 30000 else np.int32))
    # and on real geometry
    X, ids, q, M = encode(pdb_2cua, False)
    Xc, idsc, qc, Mc = collate([[X, ids, q, M]])
    save("fwd_i_v4_1_stacked_2CUA", z=run_forward(m41, Xc, idsc, qc, Mc))
    X, ids, q, M = encode(pdb_2ayo, False)   # BASELINE config 1 geometry (inputs: fwd_i_v4_0_2AYO.npz)
    Xc, idsc, qc, Mc = collate([[X, ids, q, M]])
    save("fwd_i_v4_1_stacked_2AYO", z=run_forward(m41, Xc, idsc, qc, Mc))

    # i_v3_0 (123 features) on synthetic N=512: three one-hots
    cfg30, m30 = models["i_v3_0"]
    import_reference("i_v3_0_2021-05-27_14-27")
    X, ids, q, M = synth_inputs(512, 3, n0=123)
    Xc, idsc, qc, Mc = collate([[X, ids, q, M]])
    save("fwd_i_v3_0_synth512", z=run_forward(m30, Xc, idsc, qc, Mc), seed=np.int64(3))

    # topology golden: reference extract_topology on a seeded cloud (package generator must match)
    from pesto_amd.topology import synthetic_cloud
    from src.data_encoding import extract_topology
    X = pt.from_numpy(synthetic_cloud(300, 11))
    save("topology_synth300", X=X.numpy(), ids_topk0=extract_topology(X, 64)[0].numpy().astype(np.int32))
    X = pt.from_numpy(synthetic_cloud(50, 12))   # N < 64: knn = N, self sorted to the far end
    save("topology_synth50", X=X.numpy(), ids_topk0=extract_topology(X, 64)[0].numpy().astype(np.int32))


def main_next():
    """Goldens for the SURVEY 8f rows built after the forward pass: trajectory frames (md_analysis/apply_model_md.ipynb cell 6)
    and the sigmoid / encode_bfactor post-op (apply_model.ipynb:157-166, src/structure.py:185-223)."""
    cfg, model = load_run("i_v4_0_2021-09-07_11-20")
    from src.structure import encode_bfactor
    st = parse_pdb(os.path.join(REF, "examples", "issue_19_04_2023", "2CUA_A.pdb"))
    X, ids, q, M = encode(st, False)
    _, idsc, qc, Mc = collate([[X, ids, q, M]])          # frame-0 topology + sink offset, as the MD notebook does
    rng = np.random.default_rng(21)
    frames = [X.numpy()]
    for f in range(1, 4):                                 # thermal-like displacement growing with the frame index
        frames.append((X.numpy() + rng.normal(0.0, 0.15 * f, X.shape)).astype(np.float32))
    Xf = np.stack(frames, 1)                              # the notebook's layout: [N, frames, 3]
    z = np.stack([run_forward(model, pt.from_numpy(np.ascontiguousarray(Xf[:, f])), idsc, qc, Mc) for f in range(Xf.shape[1])])
    p = pt.sigmoid(pt.from_numpy(z[0]))
    st["het_flag"] = np.array(["A"] * X.shape[0])
    bf = np.stack([encode_bfactor(dict(st), p[:, c].numpy())["bfactor"] for c in range(p.shape[1])])
    save("frames_i_v4_0_2CUA", X_traj=Xf, ids_topk=idsc.numpy().astype(np.int32), q_idx=onehot_to_idx(q, False),
         res_of_atom=res_of_atom(Mc), z=z, p0=p.numpy(), bfactor0=bf.astype(np.float32))
    # N < 64: zero-padded ids wrap to the LAST atom of each frame and the max(D) fix-up is per frame (per call in the reference)
    X, ids, q, M = synth_inputs(40, 5)
    _, idsc, qc, Mc = collate([[X, ids, q, M]])
    frames = [X.numpy()] + [(X.numpy() + rng.normal(0.0, 0.3, X.shape)).astype(np.float32) for _ in range(2)]
    Xf = np.stack(frames, 0)                              # [frames, N, 3]
    z = np.stack([run_forward(model, pt.from_numpy(Xf[f]), idsc, qc, Mc) for f in range(Xf.shape[0])])
    save("frames_i_v4_0_n40", X_frames=Xf, ids_topk=idsc.numpy().astype(np.int32), q_idx=onehot_to_idx(q, False),
         res_of_atom=res_of_atom(Mc), z=z)


PDBS_TEST_GOLDEN = ("V9_2V9T_1_B:0", "JT_1JTD_1_B:0", "WU_2WUS_1_A:0", "SJ_3SJA_3_I:1", "NV_3NVN_1_A:0")


def main_r2():
    """Round-2 goldens: (1) BASELINE config 4 - chains of pdbs_test/ through the i_v4_1 architecture (stacked weights), one
    structure per call like the reference's bulk loop (interfaceome/apply_model.py:57-82, apply_model.ipynb:139-167);
    (2) the TRAINED i_v3_1 (model/save/i_v3_1_2021-05-28_12-40/model.py:10-22 + model_ckpt.pt) on 2CUA, fp32 and fp64 - its
    states reach 4e5, beyond the f16 range of the split-MFMA path; (3) BASELINE config 3 at its stated size (i_v3_0, N=3000)."""
    import glob
    # (1) config 4
    cfg40, m40 = load_run("i_v4_0_2021-09-07_11-20")
    cfg41, Model41, _ = import_reference("i_v4_1_2021-09-07_11-21")
    m41 = Model41(cfg41).eval()
    from pesto_amd.weights import stack_layers
    sd41 = stack_layers({k: v.numpy() for k, v in m40.state_dict().items()}, cfg41, residual_scale=0.5)
    print("i_v4_1 stacked", m41.load_state_dict({k: pt.from_numpy(np.array(v)) for k, v in sd41.items()}))
    import_reference("i_v4_1_2021-09-07_11-21")
    sizes = []
    for f in sorted(glob.glob(os.path.join(REF, "pdbs_test", "*.pdb"))):
        if f.endswith(("_M.pdb", "_T.pdb")):
            continue
        st = parse_pdb(f)
        sizes.append((os.path.basename(f)[:-4], st["xyz"].shape[0], int(np.unique(st["resid"]).size)))
    save("pdbs_test_sizes", names=np.array([s[0] for s in sizes]).astype("S"), atoms=np.array([s[1] for s in sizes], np.int32),
         residues=np.array([s[2] for s in sizes], np.int32))
    for name in PDBS_TEST_GOLDEN:
        st = parse_pdb(os.path.join(REF, "pdbs_test", name + ".pdb"))
        X, ids, q, M = encode(st, False)
        Xc, idsc, qc, Mc = collate([[X, ids, q, M]])
        z = run_forward(m41, Xc, idsc, qc, Mc)
        print(f"  {name}: N={Xc.shape[0]} R={Mc.shape[1]} |z|max={np.abs(z).max():.2f}")
        save("cfg4_" + name.replace(":", "_"), X=Xc.numpy(), ids_topk=idsc.numpy().astype(np.int16), q_idx=onehot_to_idx(qc, False),
             res_of_atom=res_of_atom(Mc).astype(np.int16), z=z)
    # (2) trained i_v3_1
    cfg31, m31 = load_run("i_v3_1_2021-05-28_12-40")
    save("weights_i_v3_1", **sd_arrays(m31))
    st = parse_pdb(os.path.join(REF, "examples", "issue_19_04_2023", "2CUA_A.pdb"))
    X, ids, q, M = encode(st, True)
    Xc, idsc, qc, Mc = collate([[X, ids, q, M]])
    z32 = run_forward(m31, Xc, idsc, qc, Mc)
    with pt.no_grad():
        z64 = m31.double()(Xc.double(), idsc, qc.double(), Mc.double()).numpy()
        m31.float()
        # state magnitude the layers reach (documented in DESIGN.md)
        from src.model_operations import unpack_state_features
        q1 = m31.em(qc)
        qs, ids_s, D, R = unpack_state_features(Xc, idsc, q1)
        state = (qs, pt.zeros((qs.shape[0], 3, qs.shape[1])), ids_s, D, R)
        smax = []
        for layer in m31.sum:
            state = tuple(t.detach() for t in layer(state))
            smax.append(max(float(state[0].abs().max()), float(state[1].abs().max())))
    print(f"  i_v3_1 trained 2CUA: |z|max={np.abs(z32).max():.3e} fp32-vs-fp64 {np.abs(z32 - z64).max():.3e} state max {max(smax):.3e}")
    save("fwd_i_v3_1_2CUA", z=z32, z64=z64, state_max=np.array(smax, np.float32))   # inputs: fwd_i_v3_0_2CUA.npz (same structure, same features)
    # (3) config 3 at N=3000
    cfg30, m30 = load_run("i_v3_0_2021-05-27_14-27")
    X, ids, q, M = synth_inputs(3000, 1, n0=123)
    Xc, idsc, qc, Mc = collate([[X, ids, q, M]])
    z = run_forward(m30, Xc, idsc, qc, Mc)
    print(f"  i_v3_0 N=3000: |z|max={np.abs(z).max():.3f}")
    save("fwd_i_v3_0_synth3000", z=z, seed=np.int64(1))


def main_r3():
    """Round-3 golden: BASELINE config 4 in full - EVERY chain of pdbs_test/ (53 chains, 1,641 - 3,052 atoms) through the i_v4_1
    architecture (stacked weights), one structure per call like the reference's bulk loop (interfaceome/apply_model.py:57-82), and
    the same chains through the TRAINED i_v4_0 and i_v3_0 checkpoints (the latter with all 123 input features).
    Stored compactly: coordinates, feature indices, residue index and the reference logits. The 64-neighbour topology is recomputed by
    the tests with pesto_amd.topology.extract_topology; the generator compares it with the reference's ids (src/data_encoding.py:
    90-102) for every chain and stores the differences as a patch list. They can only be exact fp32 distance TIES, which torch.topk
    orders arbitrarily (checked here: same distance, same neighbour set per row) - with the patches applied the tests feed the
    reference's own ids. Also checked here for every file: the native reader + preprocessing + encoder (libpesto_io.so) give exactly
    the coordinates, features and residue map the reference's encode_structure / encode_features give."""
    import glob
    import pickle
    cache = "/tmp/pesto_r3_cache"
    os.makedirs(cache, exist_ok=True)
    cfg40, m40 = load_run("i_v4_0_2021-09-07_11-20")
    cfg41, Model41, _ = import_reference("i_v4_1_2021-09-07_11-21")
    m41 = Model41(cfg41).eval()
    from pesto_amd.weights import stack_layers
    from pesto_amd import topology
    from pesto_amd.structure_io import Structure
    sd41 = stack_layers({k: v.numpy() for k, v in m40.state_dict().items()}, cfg41, residual_scale=0.5)
    print("i_v4_1 stacked", m41.load_state_dict({k: pt.from_numpy(np.array(v)) for k, v in sd41.items()}))
    cfg30, m30 = load_run("i_v3_0_2021-05-27_14-27")
    import_reference("i_v4_1_2021-09-07_11-21")
    models = (("i_v4_1", m41, False), ("i_v4_0", m40, False), ("i_v3_0", m30, True))

    def cached(name, tag, fn):
        cf = os.path.join(cache, f"{name}.{tag}.pkl")      # (a reference forward takes up to a minute per chain: its logits are cached in /tmp)
        if os.path.exists(cf):
            return pickle.load(open(cf, "rb"))
        z = fn().astype(np.float32)
        pickle.dump(z, open(cf, "wb"))
        return z

    names, Xs, qs, q3s, roas, patches = [], [], [], [], [], []
    zs = {tag: [] for tag, _, _ in models}
    for f in sorted(glob.glob(os.path.join(REF, "pdbs_test", "*.pdb"))):
        if f.endswith(("_M.pdb", "_T.pdb")):
            continue
        name = os.path.basename(f)[:-4]
        st = parse_pdb(f)
        X, ids, q, M = encode(st, False)
        _, _, q_all, _ = encode(st, True)
        Xc, idsc, qc, Mc = collate([[X, ids, q, M]])
        _, _, qc_all, _ = collate([[X, ids, q_all, M]])
        Xn = Xc.numpy().astype(np.float32)
        qi, qi3, roa = onehot_to_idx(qc, False).astype(np.uint8), onehot_to_idx(qc_all, True).astype(np.uint8), res_of_atom(Mc).astype(np.int16)
        # native structure I/O on the same file
        sn = Structure.read_pdb(f)
        sn.preprocess()
        Xio, q0io, roaio, Rio = sn.encode(30)
        assert np.array_equal(Xio, Xn) and np.array_equal(q0io.argmax(1), qi[:, 0]) and np.array_equal(roaio, roa) and Rio == Mc.shape[1], name
        assert np.array_equal(sn.encode(123)[1], qc_all.numpy()), name
        mine = np.asarray(topology.extract_topology(Xn, 64)).astype(np.int64)
        ref0 = idsc.numpy().astype(np.int64) - 1
        pch = []
        diff = np.argwhere(mine != ref0)
        if len(diff):
            D = pt.norm(Xc.unsqueeze(0) - Xc.unsqueeze(1), dim=2).numpy()      # the reference's own distance matrix (data_encoding.py:90)
            for r, c in diff:
                assert D[r, mine[r, c]] == D[r, ref0[r, c]] and sorted(mine[r]) == sorted(ref0[r]), (name, r, c)   # a tie, nothing else
                pch.append((int(r), int(c), int(ref0[r, c])))
        for tag, mdl, all_f in models:
            zs[tag].append(cached(name, tag, lambda: run_forward(mdl, Xc, idsc, qc_all if all_f else qc, Mc)))
        print(f"  {name}: N={Xn.shape[0]} R={Mc.shape[1]} tie patches {len(pch)} |z|max " +
              " ".join(f"{tag} {np.abs(zs[tag][-1]).max():.2f}" for tag, _, _ in models), flush=True)
        patches += [(len(names), r, c, v) for r, c, v in pch]
        names.append(name); Xs.append(Xn); qs.append(qi); q3s.append(qi3); roas.append(roa)
    pa = np.array(patches, np.int32).reshape(-1, 4)
    save("cfg4_all53", names=np.array(names).astype("S"), atom_offsets=np.cumsum([0] + [x.shape[0] for x in Xs]).astype(np.int32),
         res_offsets=np.cumsum([0] + [z.shape[0] for z in zs["i_v4_1"]]).astype(np.int32), X=np.concatenate(Xs, 0), q_idx=np.concatenate(qs, 0),
         q_idx3=np.concatenate(q3s, 0), res_of_atom=np.concatenate(roas, 0), z=np.concatenate(zs["i_v4_1"], 0),
         z_i_v4_0=np.concatenate(zs["i_v4_0"], 0), z_i_v3_0=np.concatenate(zs["i_v3_0"], 0), tie_patches=pa)


EXAMPLE_COMPLEXES = ("endonuclease/1ZNS", "lipids/7KHT_lipid", "dna_rna/3IVK", "dna_rna/1H9D", "lipids/6O1T", "lipids/6XRU", "channel/6Y5B")


def main_r3_examples():
    """Round-3 golden: multi-chain complexes of the reference's examples/ with DNA / RNA, lipids, ions and ligands (955 - 15,635 atoms;
    two of them beyond the 4,096 atoms where the host topology switches to the k-d tree and the GPU k-NN to its cell grid) through the
    TRAINED i_v4_0 and i_v3_0 checkpoints, as apply_model.ipynb cell 6 does: read -> preprocessing chain -> encode -> topology ->
    forward. The native reader feeds the reference's own preprocessing / encoding functions (its reader needs gemmi; the native one is
    pinned against every file of examples/ by sweep_examples.py). Stored: coordinates, feature indices, residue map, reference logits,
    tie patch list (see main_r3)."""
    import pickle
    cache = "/tmp/pesto_r3_cache"
    os.makedirs(cache, exist_ok=True)
    cfg40, m40 = load_run("i_v4_0_2021-09-07_11-20")
    cfg30, m30 = load_run("i_v3_0_2021-05-27_14-27")
    sys.path = [REF] + [p for p in sys.path if "/model/save/" not in p and p != REF]
    for m in [m for m in sys.modules if m == "src" or m.startswith("src.")]:
        sys.modules.pop(m)
    from src.structure import (clean_structure, tag_hetatm_chains, split_by_chain, filter_non_atomic_subunits,
                               remove_duplicate_tagged_subunits, concatenate_chains)
    from src.data_encoding import encode_structure, encode_features, extract_topology
    from pesto_amd import topology
    from pesto_amd.structure_io import Structure
    names, Xs, q3s, roas, patches = [], [], [], [], []
    zs = {"i_v4_0": [], "i_v3_0": []}
    for rel in EXAMPLE_COMPLEXES:
        name = os.path.basename(rel)
        st = Structure.read_pdb(os.path.join(REF, "examples", rel + ".pdb")).to_dict()
        st["resid"] = st["resid"].astype(np.int32)
        s = concatenate_chains(remove_duplicate_tagged_subunits(filter_non_atomic_subunits(split_by_chain(tag_hetatm_chains(clean_structure(st))))))
        X, M = encode_structure(s)
        qs = encode_features(s)
        ids = extract_topology(X, 64)[0]
        Xc, idsc, qc, Mc = collate([[X, ids, qs[0], M]])
        _, _, qc_all, _ = collate([[X, ids, pt.cat(qs, dim=1), M]])
        Xn = Xc.numpy().astype(np.float32)
        mine = np.asarray(topology.extract_topology(Xn, 64)).astype(np.int64)
        ref0 = idsc.numpy().astype(np.int64) - 1
        pch = []
        for r, c in np.argwhere(mine != ref0):
            d = pt.norm(Xc - Xc[r], dim=1).numpy()          # this row of the reference's distance matrix
            assert d[mine[r, c]] == d[ref0[r, c]], (name, r, c)   # a tie, nothing else (two slots swapped, or the 64th / 65th candidate)
            pch.append((int(r), int(c), int(ref0[r, c])))
        for tag, mdl, q_in in (("i_v4_0", m40, qc), ("i_v3_0", m30, qc_all)):
            cf = os.path.join(cache, f"ex_{name}.{tag}.pkl")
            if os.path.exists(cf):
                z = pickle.load(open(cf, "rb"))
            else:
                z = run_forward(mdl, Xc, idsc, q_in, Mc).astype(np.float32)
                pickle.dump(z, open(cf, "wb"))
            zs[tag].append(z)
        print(f"  {name}: N={Xn.shape[0]} R={Mc.shape[1]} tie patches {len(pch)} |z|max i_v4_0 {np.abs(zs['i_v4_0'][-1]).max():.2f} "
              f"i_v3_0 {np.abs(zs['i_v3_0'][-1]).max():.2f}", flush=True)
        patches += [(len(names), r, c, v) for r, c, v in pch]
        names.append(name); Xs.append(Xn); q3s.append(onehot_to_idx(qc_all, True).astype(np.uint8)); roas.append(res_of_atom(Mc).astype(np.int16))
    save("examples_complexes", names=np.array(names).astype("S"), atom_offsets=np.cumsum([0] + [x.shape[0] for x in Xs]).astype(np.int32),
         res_offsets=np.cumsum([0] + [z.shape[0] for z in zs["i_v4_0"]]).astype(np.int32), X=np.concatenate(Xs, 0), q_idx3=np.concatenate(q3s, 0),
         res_of_atom=np.concatenate(roas, 0), z_i_v4_0=np.concatenate(zs["i_v4_0"], 0), z_i_v3_0=np.concatenate(zs["i_v3_0"], 0),
         tie_patches=np.array(patches, np.int32).reshape(-1, 4))


def main_r3_frames(family="1JTG_uL"):
    """Round-3 golden for the trajectory path on REAL conformations: the MD cluster representatives of one molecule from the reference's
    md_analysis/pdbs_clusters/ (29 conformations of 1JTG's ligand-side subunit), run as md_analysis/apply_model_md.ipynb cell 6 does -
    topology, features and residue map of frame 0 for every frame, one forward per frame - with the trained i_v4_0 checkpoint."""
    import glob
    import pickle
    cache = "/tmp/pesto_r3_cache"
    os.makedirs(cache, exist_ok=True)
    cfg40, m40 = load_run("i_v4_0_2021-09-07_11-20")
    sys.path = [REF] + [p for p in sys.path if "/model/save/" not in p and p != REF]
    for m in [m for m in sys.modules if m == "src" or m.startswith("src.")]:
        sys.modules.pop(m)
    from src.structure import (clean_structure, tag_hetatm_chains, split_by_chain, filter_non_atomic_subunits,
                               remove_duplicate_tagged_subunits, concatenate_chains)
    from src.data_encoding import encode_structure, encode_features, extract_topology
    from pesto_amd import topology
    from pesto_amd.structure_io import Structure
    files = sorted(glob.glob(os.path.join(REF, "md_analysis", "pdbs_clusters", family + "_*_AUC*.pdb")),
                   key=lambda f: int(os.path.basename(f).split("_")[2]))
    frames, first = [], None
    for f in files:
        st = Structure.read_pdb(f).to_dict()
        st["resid"] = st["resid"].astype(np.int32)
        s = concatenate_chains(remove_duplicate_tagged_subunits(filter_non_atomic_subunits(split_by_chain(tag_hetatm_chains(clean_structure(st))))))
        if first is None:
            first = s
        assert all(np.array_equal(s[k], first[k]) for k in ("name", "element", "resname", "resid", "chain_name")), f     # the same atoms in every file
        frames.append(s["xyz"].astype(np.float32))
    X0, M = encode_structure(first)
    qs = encode_features(first)
    ids = extract_topology(X0, 64)[0]
    _, idsc, qc, Mc = collate([[X0, ids, qs[0], M]])
    Xf = np.stack(frames, 0)
    mine = np.asarray(topology.extract_topology(Xf[0], 64)).astype(np.int64)
    ref0 = idsc.numpy().astype(np.int64) - 1
    pch = []
    for r, c in np.argwhere(mine != ref0):
        d = pt.norm(X0 - X0[r], dim=1).numpy()
        assert d[mine[r, c]] == d[ref0[r, c]], (family, r, c)
        pch.append((int(r), int(c), int(ref0[r, c])))
    cf = os.path.join(cache, f"frames_{family}.pkl")
    if os.path.exists(cf):
        z = pickle.load(open(cf, "rb"))
    else:
        z = np.stack([run_forward(m40, pt.from_numpy(Xf[i]), idsc, qc, Mc) for i in range(Xf.shape[0])], 0).astype(np.float32)
        pickle.dump(z, open(cf, "wb"))
    print(f"  {family}: {Xf.shape[0]} frames, N={Xf.shape[1]} R={Mc.shape[1]} tie patches {len(pch)} |z|max {np.abs(z).max():.2f} "
          f"max frame-to-frame |dz| {np.abs(z[1:] - z[:-1]).max():.2f}")
    save("frames_md_" + family, X_frames=Xf, q_idx=onehot_to_idx(qc, False).astype(np.uint8), res_of_atom=res_of_atom(Mc).astype(np.int16), z=z,
         tie_patches=np.array(pch, np.int32).reshape(-1, 3))


def _pdb_line(rec, serial, name, alt, resname, chain, resnum, icode, xyz, element, occ=1.0, b=20.0):
    name4 = name if len(name) == 4 else " " + name.ljust(3)
    return "%-6s%5d %4s%1s%3s %1s%4d%1s   %8.3f%8.3f%8.3f%6.2f%6.2f          %2s  " % (
        rec, serial, name4, alt, resname, chain, resnum, icode, xyz[0], xyz[1], xyz[2], occ, b, element.upper().rjust(2))


def synthetic_pdb_text():
    """A small two-model file written for this test suite: insertion code, alternate locations (also across models), water,
    heavy water, H and D atoms, a CA-only chain, ions and ligands, duplicated ligands between the models."""
    rng = np.random.default_rng(5)
    lines, serial = ["HEADER    SYNTHETIC TEST STRUCTURE", "REMARK   1 two models"], 0
    for mid in (1, 2):
        lines.append("MODEL     %4d" % mid)
        shift = np.array([0.0, 0.0, 12.0 * (mid - 1)])

        def add(rec, name, alt, resname, chain, num, icode, xyz, el):
            nonlocal serial
            serial += 1
            lines.append(_pdb_line(rec, serial, name, alt, resname, chain, num, icode, np.asarray(xyz) + shift, el))
        base = np.array([10.0, 5.0, 3.0])
        for k, (rn, num, ic, names) in enumerate([("ALA", 1, " ", ["N", "CA", "C", "O", "CB", "H", "HA"]),
                                                   ("GLY", 1, "A", ["N", "CA", "C", "O", "D"]),
                                                   ("SER", 2, " ", ["N", "CA", "C", "O", "CB", "OG"]),
                                                   ("MSE", 3, " ", ["N", "CA", "C", "O", "SE"])]):
            for j, nm in enumerate(names):
                xyz = base + np.array([3.8 * k, 1.3 * j, 0.4 * j]) + rng.normal(0, 0.2, 3)
                el = {"SE": "SE", "H": "H", "HA": "H", "D": "D"}.get(nm, nm[0])
                if rn == "SER" and nm in ("CB", "OG"):      # two alternate locations; the reference keeps the first seen key
                    add("ATOM", nm, "A", rn, "A", num, ic, xyz, el)
                    add("ATOM", nm, "B", rn, "A", num, ic, xyz + 0.7, el)
                elif rn == "MSE":
                    add("HETATM", nm, " ", rn, "A", num, ic, xyz, el)
                else:
                    add("ATOM", nm, " ", rn, "A", num, ic, xyz, el)
        lines.append("TER")
        for k in range(3):                                   # CA-only chain: one atom per residue -> filtered out
            add("ATOM", "CA", " ", "LYS", "B", 10 + k, " ", base + [3.8 * k, -8.0, 0.0], "C")
        lines.append("TER")
        lig = np.array([25.0, 9.0, 4.0]) - shift              # ligands at (almost) the same place in both models -> duplicates
        add("HETATM", "ZN", " ", " ZN", "A", 201, " ", lig, "ZN")
        for j, (nm, el) in enumerate([("S", "S"), ("O1", "O"), ("O2", "O"), ("O3", "O"), ("O4", "O")]):
            add("HETATM", nm, " ", "SO4", "A", 202, " ", lig + [4.0 + 1.2 * j, 0.3 * j, 0.05 * (mid - 1)], el)
        add("HETATM", "FE", " ", "HEM", "C", 301, " ", base + [0.0, 9.0, 12.0 * (mid - 1) * 0 + 3.0 * mid], "FE")   # not duplicates
        add("HETATM", "O", " ", "HOH", "A", 401, " ", base + [1.0, 1.0, 8.0], "O")
        add("HETATM", "O", " ", "DOD", "A", 402, " ", base + [2.0, 1.0, 8.0], "O")
        lines.append("ENDMDL")
    lines.append("END")
    return "\n".join(lines) + "\n"


def main_io():
    """Goldens for the native structure I/O: the reference's own preprocessing / encoding / writing
    functions applied to the dict the native reader produces (the reference's reader needs gemmi, absent here - the reader is
    pinned separately by the reference's examples/*.pdb -> *_i0.pdb pairs copied to tests/golden/pdb/)."""
    import gzip
    import shutil
    import tempfile
    import_reference("i_v4_0_2021-09-07_11-20")     # installs the gemmi stub
    # the repository's own src/ (what apply_model.ipynb imports), not the older snapshot stored inside the run directory
    sys.path = [REF] + [p for p in sys.path if "/model/save/" not in p and p != REF]
    for m in [m for m in sys.modules if m == "src" or m.startswith("src.")]:
        sys.modules.pop(m)
    from src.structure import (clean_structure, tag_hetatm_chains, split_by_chain, filter_non_atomic_subunits,
                               remove_duplicate_tagged_subunits, concatenate_chains, encode_bfactor)
    from src.data_encoding import encode_structure, encode_features
    from src.structure_io import save_pdb
    from pesto_amd.structure_io import Structure
    os.makedirs(os.path.join(OUT, "pdb"), exist_ok=True)
    ex = os.path.join(REF, "examples")
    cases = {"synthetic": synthetic_pdb_text()}
    for rel in ("lipids/7KHT_lipid", "double/1thf_D", "lipids/6I9F", "endonuclease/1ZNS_ion"):
        name = os.path.basename(rel)
        for suffix in (".pdb", "_i0.pdb"):               # reference DATA files: an input and the output the reference saved for it
            src = os.path.join(ex, rel + suffix)
            if os.path.exists(src):
                with open(src, "rb") as fi, gzip.GzipFile(os.path.join(OUT, "pdb", name + suffix + ".gz"), "wb", mtime=0) as fo:
                    shutil.copyfileobj(fi, fo)
        cases[name] = open(os.path.join(ex, rel + ".pdb")).read()
    rng = np.random.default_rng(9)
    for name, text in cases.items():
        st = Structure.parse_pdb(text).to_dict()
        st["resid"] = st["resid"].astype(np.int32)
        out = {"pdb_text": np.frombuffer(text.encode(), dtype=np.uint8)}
        s = clean_structure({k: v.copy() for k, v in st.items()})
        out.update(clean_resid=s["resid"], clean_chain=s["chain_name"].astype("S"), clean_name=s["name"].astype("S"))
        s = tag_hetatm_chains(s)
        out["tag_chain"] = s["chain_name"].astype("S")
        sub = split_by_chain(s)
        out["split_keys"] = np.array(list(sub)).astype("S")
        sub = filter_non_atomic_subunits(sub)
        out["filter_keys"] = np.array(list(sub)).astype("S")
        sub = remove_duplicate_tagged_subunits(sub)
        out["dedup_keys"] = np.array(list(sub)).astype("S")
        s = concatenate_chains(sub)
        for k in ("xyz", "resid"):
            out["final_" + k] = s[k]
        for k in ("name", "element", "resname", "het_flag", "chain_name"):
            out["final_" + k] = s[k].astype("S")
        X, M = encode_structure(s)
        qe, qr, qn = encode_features(s)
        out["M_col"] = M.numpy().argmax(1).astype(np.int32)
        out["n_res"] = np.int64(M.shape[1])
        out["q_idx"] = np.stack([qe.numpy().argmax(1), qr.numpy().argmax(1), qn.numpy().argmax(1)], 1).astype(np.int16)
        p = rng.uniform(0, 1, M.shape[1]).astype(np.float32)       # a per-residue prediction
        s = encode_bfactor(s, p)
        with tempfile.TemporaryDirectory() as tmp:
            save_pdb(split_by_chain(s), os.path.join(tmp, "o.pdb"))
            out["saved_text"] = np.frombuffer(open(os.path.join(tmp, "o.pdb"), "rb").read(), dtype=np.uint8)
        out["p_res"] = p
        out["bfactor"] = s["bfactor"].astype(np.float32)
        save("io_" + name, **out)


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    if "--next" in sys.argv:      # only the 8f-row goldens (the others are unchanged)
        main_next()
    elif "--io" in sys.argv:
        main_io()
    elif "--r2" in sys.argv:      # round-2 additions only
        main_r2()
    elif "--r3" in sys.argv:      # round-3 addition only (all 53 config-4 chains)
        main_r3()
    elif "--r3-examples" in sys.argv:
        main_r3_examples()
    elif "--r3-frames" in sys.argv:
        main_r3_frames()
    else:
        main()
        main_next()
        main_io()
        main_r2()
        main_r3()
        main_r3_examples()
        main_r3_frames()
