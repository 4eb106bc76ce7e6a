import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no built libraries (they are git-ignored): build them once, in-tree (hipcc cross-compiles without a GPU)
    from pesto_amd import _lib, structure_io
    if not (os.path.exists(_lib.LIB_PATH) and os.path.exists(structure_io.LIB_PATH)):
        from pesto_amd.csrc import build as native_build
        native_build.build(verbose=False)


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def onehot(idx, n0):
    """q_idx [N, 1 or 3] -> dense one-hot float32 [N, n0] (element | resname | atom-name blocks)."""
    idx = np.asarray(idx, dtype=np.int64)
    q = np.zeros((idx.shape[0], n0), np.float32)
    offs = [0, 30, 59]
    for c in range(idx.shape[1]):
        q[np.arange(idx.shape[0]), offs[c] + idx[:, c]] = 1.0
    return q


_weights_cache = {}


def weights(tag):
    """state_dict (numpy) for a fixture tag: real i_v4_0/i_v3_0/i_v3_1, or i_v4_1 = stacked i_v4_0."""
    if tag not in _weights_cache:
        from pesto_amd.config import CONFIGS
        from pesto_amd.weights import stack_layers
        if tag == "i_v4_1":
            _weights_cache[tag] = stack_layers(weights("i_v4_0"), CONFIGS["i_v4_1"], 0.5)
        elif tag == "i_v3_1_trained":  # the reference's trained i_v3_1 (model/save/i_v3_1_2021-05-28_12-40/model_ckpt.pt) as arrays
            d = golden("weights_i_v3_1")
            _weights_cache[tag] = {k: d[k] for k in d.files}
        elif tag == "i_v3_1":  # hybrid: own em/dm + i_v3_0 sum/spl (see tests/golden/make_golden.py)
            sd = dict(weights("i_v3_0"))
            d = golden("weights_i_v3_1_emdm")
            sd = {k: v for k, v in sd.items() if not k.startswith(("em.", "dm."))}
            sd.update({k: d[k] for k in d.files})
            _weights_cache[tag] = sd
        else:
            d = golden("weights_" + tag)
            _weights_cache[tag] = {k: d[k] for k in d.files}
    return _weights_cache[tag]


# BASELINE config 4: chains of the reference's pdbs_test/ set whose reference outputs (i_v4_1 architecture, stacked weights) are
# committed as cfg4_<name>.npz; pdbs_test_sizes.npz holds the atom / residue counts of all 53 chains
CFG4_CHAINS = ("V9_2V9T_1_B_0", "JT_1JTD_1_B_0", "WU_2WUS_1_A_0", "SJ_3SJA_3_I_1", "NV_3NVN_1_A_0")


def cfg4_structure(name):
    """(X, ids_topk0 [N,64] 0-based, q0, M, z_ref) of one config-4 chain, in the per-structure contract of collate_batch_features."""
    g = golden("cfg4_" + name)
    roa = g["res_of_atom"].astype(np.int32)
    M = np.zeros((roa.size, int(roa.max()) + 1), np.float32)
    M[np.arange(roa.size), roa] = 1.0
    return g["X"], g["ids_topk"].astype(np.int32) - 1, onehot(g["q_idx"], 30), M, g["z"]


def cfg4_all53(only=None):
    """Every chain of pdbs_test/ (BASELINE config 4 in full): list of dicts name, X, ids0 (the reference's 0-based ids), ids0_host
    (pesto_amd.topology.extract_topology), q0 / q0_all (30 / 123 input features), res_of_atom, R and the reference logits: z (i_v4_1
    architecture, stacked weights), z_i_v4_0 and z_i_v3_0 (the TRAINED checkpoints). The fixture holds coordinates, feature /
    residue indices, the reference logits and a patch list: the topology is recomputed on the host and the generator stored where
    the reference's ids differ from it - exact fp32 distance ties only, which torch.topk orders arbitrarily."""
    from pesto_amd.topology import extract_topology
    g = golden("cfg4_all53")
    ao, ro, pa = g["atom_offsets"], g["res_offsets"], g["tie_patches"]
    out = []
    for i, name in enumerate(g["names"]):
        if only is not None and name.decode() not in only:
            continue
        X = g["X"][ao[i]:ao[i + 1]]
        host = np.asarray(extract_topology(X, 64)).astype(np.int32)
        ids0 = host.copy()
        for _, r, c, v in pa[pa[:, 0] == i]:
            ids0[r, c] = v
        out.append(dict(name=name.decode(), X=X, ids0=ids0, ids0_host=host, q0=onehot(g["q_idx"][ao[i]:ao[i + 1]], 30),
                        q0_all=onehot(g["q_idx3"][ao[i]:ao[i + 1]], 123),      # element | residue | atom-name blocks (i_v3_*)
                        res_of_atom=g["res_of_atom"][ao[i]:ao[i + 1]].astype(np.int32), R=int(ro[i + 1] - ro[i]), z=g["z"][ro[i]:ro[i + 1]],
                        z_i_v4_0=g["z_i_v4_0"][ro[i]:ro[i + 1]], z_i_v3_0=g["z_i_v3_0"][ro[i]:ro[i + 1]]))
    return out


def example_complexes(only=None):
    """Multi-chain complexes of the reference's examples/ (protein + DNA / RNA / lipid / ion / ligand chains, 955 - 15,635 atoms) with
    the logits of the TRAINED i_v4_0 and i_v3_0 checkpoints: list of dicts like cfg4_all53 (ids0 = the reference's neighbour ids:
    host topology + the fixture's tie patches)."""
    from pesto_amd.topology import extract_topology
    g = golden("examples_complexes")
    ao, ro, pa = g["atom_offsets"], g["res_offsets"], g["tie_patches"]
    out = []
    for i, name in enumerate(g["names"]):
        if only is not None and name.decode() not in only:
            continue
        X = g["X"][ao[i]:ao[i + 1]]
        host = np.asarray(extract_topology(X, 64)).astype(np.int32)
        ids0 = host.copy()
        for _, r, c, v in pa[pa[:, 0] == i]:
            ids0[r, c] = v
        q3 = g["q_idx3"][ao[i]:ao[i + 1]]
        out.append(dict(name=name.decode(), X=X, ids0=ids0, ids0_host=host, q0=onehot(q3[:, :1], 30), q0_all=onehot(q3, 123),
                        res_of_atom=g["res_of_atom"][ao[i]:ao[i + 1]].astype(np.int32), R=int(ro[i + 1] - ro[i]),
                        z_i_v4_0=g["z_i_v4_0"][ro[i]:ro[i + 1]], z_i_v3_0=g["z_i_v3_0"][ro[i]:ro[i + 1]]))
    return out


def md_frames(name="1JTG_uL"):
    """Real MD conformations of one molecule (cluster representatives from the reference's md_analysis/pdbs_clusters/) with the logits of
    the reference's per-frame loop (frame-0 topology for every frame, trained i_v4_0): X_frames [F,N,3], ids (1-based, the reference's:
    host topology of frame 0 + tie patches), q0, res_of_atom, R, z [F,R,5]."""
    from pesto_amd.topology import extract_topology
    g = golden("frames_md_" + name)
    ids0 = np.asarray(extract_topology(g["X_frames"][0], 64)).astype(np.int64)
    for r, c, v in g["tie_patches"]:
        ids0[r, c] = v
    return dict(X_frames=g["X_frames"], ids=ids0 + 1, q0=onehot(g["q_idx"], 30), res_of_atom=g["res_of_atom"].astype(np.int32),
                R=int(g["z"].shape[1]), z=g["z"])


@pytest.fixture(scope="session")
def gpu_available():
    import torch
    return torch.cuda.is_available()
