"""Input contract of the hot path: what ``Model.forward(X, ids_topk, q, M)`` is fed.

The reference produces these tensors with src/data_encoding.py:61-102 (encode_structure,
encode_features, extract_topology) and src/dataset.py:91-112 (collate_batch_features); the
north-star keeps that preprocessing unchanged.  This module is the build's own counterpart, needed
for synthetic clouds (bench, tests) and for the sharded driver:

  extract_topology(X, k)        exact k nearest neighbours, ascending distance, the reference's
                                "D < 1e-2 -> D + max(D)" rule (self / coincident atoms go last)
  collate_batch_features(batch) concat structures, 1-based ids with per-structure offset
                                (0 = sink / padding), block-diagonal residue mask
  mask_to_segments(M)           [N, R] 0/1 mask -> res_of_atom[N] int32 (what the kernels use)
  synthetic_structure(n, seed)  uniform cloud at protein heavy-atom density

Everything here is numpy (torch tensors are accepted and returned where given).
"""
import numpy as np

try:  # scipy is present in the image; only needed for N > 4096
    from scipy.spatial import cKDTree
except Exception:  # pragma: no cover
    cKDTree = None


def _np(a):
    if hasattr(a, "detach"):
        return a.detach().cpu().numpy()
    return np.asarray(a)


def _like(ref, a):
    """Return ``a`` as the same kind of object as ``ref`` (torch tensor or numpy)."""
    if hasattr(ref, "detach"):
        import torch
        return torch.from_numpy(np.ascontiguousarray(a)).to(ref.device)
    return a


# --------------------------------------------------------------------------- k-NN topology
def _norm_xyz(R):
    """float32 ||R|| over the last (xyz) axis exactly as torch.norm computes it in the reference's build: the scalar reduction
    acc = acc + v * v contracted into an FMA chain, sqrt(fma(z, z, fma(y, y, x * x))) - bit-identical to the reference's distance
    matrix on 555,000 pairs of a pdbs_test chain, where separately rounded products agree on 89 % only (and a one-ulp difference
    reorders two neighbours in about one row of 40,000). The FMAs are emulated in float64: a product of two float32 is exact there and the
    sum is rounded once (_fma_round_f32)."""
    x, y, z = (R[..., c].astype(np.float64) for c in range(3))
    t = (x * x).astype(np.float32).astype(np.float64)
    t = _fma_round_f32(y * y, t).astype(np.float64)
    return np.sqrt(_fma_round_f32(z * z, t))


def _fma_round_f32(p, c):
    """float32(p + c) with ONE rounding, p an exact float64 product of two float32, c a float32 held in float64: the float64 sum is
    rounded to odd first (TwoSum residual != 0 and an even last bit -> the neighbour towards the residual), after which the rounding to
    float32 cannot double-round - a plain float64 add followed by the cast can, when the sum falls exactly on a float32 tie."""
    s = p + c
    bb = s - p
    e = (p - (s - bb)) + (c - bb)                     # exact residual of the float64 addition
    even = (s.view(np.int64) & 1) == 0
    nudge = (e != 0) & even & np.isfinite(s)
    s = np.where(nudge, np.nextafter(s, np.where(e > 0, np.inf, -np.inf)), s)
    return s.astype(np.float32)


def _topology_dense(X, knn):
    """Dense O(N^2) restatement, float32 like the reference (src/data_encoding.py:87-99)."""
    R = X[None, :, :] - X[:, None, :]
    D = _norm_xyz(R)
    D = D + np.max(D) * (D < 1e-2).astype(np.float32)
    ids = np.argsort(D, axis=1, kind="stable")[:, :knn]
    return ids.astype(np.int64)


def _topology_tree(X, knn):
    """Same contract through a k-d tree (exact), for sizes where [N,N] does not fit comfortably."""
    n = X.shape[0]
    X64 = X.astype(np.float64)
    tree = cKDTree(X64)
    extra = 8
    while True:
        kq = min(n, knn + extra)
        _, idx = tree.query(X64, k=kq)
        idx = np.sort(idx.reshape(n, kq), axis=1)      # candidates in index order: the stable sorts below then break exact fp32
                                                       # distance ties by index, like the dense path and k_knn_collate (the tree's
                                                       # own order is by float64 distance)
        # float32 distances computed the reference's way, then its masking rule
        R = X[idx] - X[:, None, :]
        D = _norm_xyz(R)
        close = D < 1e-2
        if kq < n and np.any(np.sum(~close, axis=1) < knn):
            extra *= 2  # many coincident atoms: widen the query
            continue
        break
    if kq < n:
        # masked entries would sort after every real neighbour: drop them, keep ascending order
        order = np.argsort(np.where(close, np.float32(np.inf), D), axis=1, kind="stable")[:, :knn]
        return np.take_along_axis(idx, order, axis=1).astype(np.int64)
    return _topology_dense(X, knn)


def extract_topology(X, num_nn=64):
    """ids_topk [N, min(num_nn, N)] int64, 0-based, ascending distance (src/data_encoding.py:87-102).

    Only the index tensor is returned (the reference also returns D/R, which callers discard:
    apply_model.ipynb:149, profiling.py:92)."""
    Xn = np.ascontiguousarray(_np(X), dtype=np.float32)
    n = Xn.shape[0]
    knn = min(num_nn, n)
    if n <= 4096 or cKDTree is None:
        ids = _topology_dense(Xn, knn)
    else:
        ids = _topology_tree(Xn, knn)
    return _like(X, ids)


# --------------------------------------------------------------------------- collation
def collate_batch_features(batch_data, max_num_nn=64):
    """[[X, ids_topk0, q, M], ...] -> (X, ids_topk, q, M) with the reference's contract
    (src/dataset.py:91-112): ids become 1-based with a per-structure offset, zero-padded to
    ``max_num_nn`` columns (0 = sink); M is block-diagonal float32."""
    ref = batch_data[0][0]
    Xs = [np.asarray(_np(d[0]), dtype=np.float32) for d in batch_data]
    qs = [np.asarray(_np(d[2]), dtype=np.float32) for d in batch_data]
    n_tot = sum(x.shape[0] for x in Xs)
    r_tot = sum(_np(d[3]).shape[1] for d in batch_data)
    ids_topk = np.zeros((n_tot, max_num_nn), dtype=np.int64)
    M = np.zeros((n_tot, r_tot), dtype=np.float32)
    ix0 = iy0 = 0
    for d in batch_data:
        ids = _np(d[1]).astype(np.int64)
        Mi = _np(d[3])
        n, r = Mi.shape
        if ids.shape[1] > max_num_nn:
            raise ValueError("more neighbour columns than max_num_nn")
        ids_topk[ix0:ix0 + n, :ids.shape[1]] = ids + ix0 + 1
        M[ix0:ix0 + n, iy0:iy0 + r] = Mi
        ix0 += n
        iy0 += r
    return (_like(ref, np.concatenate(Xs, 0)), _like(ref, ids_topk), _like(ref, np.concatenate(qs, 0)), _like(ref, M))


def mask_to_segments(M):
    """[N, R] 0/1 residue mask -> (res_of_atom int32 [N], R).  Every atom must belong to exactly one
    residue and every residue must be non-empty; the reference's dense softmax (model_operations.py:199-205)
    silently produces uniform weights / NaN otherwise, the C ABI rejects it instead."""
    if hasattr(M, "detach") and M.is_cuda:
        import torch
        Mb = M > 0.5
        cnt = Mb.sum(1)
        if not bool((cnt == 1).all()):
            raise ValueError("M: every atom must belong to exactly one residue")
        if not bool(Mb.any(0).all()):
            raise ValueError("M: empty residue column")
        return torch.argmax(Mb.to(torch.int8), dim=1).to(torch.int32), int(M.shape[1])
    Mh = _np(M)
    if Mh.ndim == 2 and Mh.dtype in (np.float32, np.bool_, np.uint8) and Mh.flags.c_contiguous and Mh.size:
        # one native pass over the rows (libpesto_io.so, host only) instead of four numpy passes over the dense mask: this reduction is
        # the largest host cost per structure of the bulk path when callers hand over the reference's dense M (profiles/r04_host_packing.json)
        try:
            from . import structure_io
            lib = structure_io.load()
        except Exception:
            lib = None
        if lib is not None:
            roa = np.empty(Mh.shape[0], np.int32)
            # bool / uint8: the mask as encode_structure returns it (src/data_encoding.py:61-75) - a quarter of the float mask's bytes
            if lib.pesto_io_mask_to_segments_any(Mh.ctypes.data, Mh.dtype.itemsize, Mh.shape[0], Mh.shape[1], roa.ctypes.data) != 0:
                raise ValueError(lib.pesto_io_last_error().decode())
            return roa, int(Mh.shape[1])
    Mn = Mh > 0.5
    if not np.all(Mn.sum(1) == 1):
        raise ValueError("M: every atom must belong to exactly one residue")
    if not np.all(Mn.any(0)):
        raise ValueError("M: empty residue column")
    return np.argmax(Mn, axis=1).astype(np.int32), int(Mn.shape[1])


# --------------------------------------------------------------------------- synthetic clouds
ELEMENT_P = np.array([0.63, 0.19, 0.17, 0.01] + [0.0] * 26)  # C, O, N, S (std_elements order)


def synthetic_cloud(n, seed=1, density=0.05, min_sep=0.5):
    """n points uniform in a cube at ``density`` atoms/A^3, sequentially rejecting points closer
    than ``min_sep`` to an accepted one (keeps timing runs off the D < 1e-2 path). float32 [n, 3]."""
    rng = np.random.default_rng(seed)
    side = (n / density) ** (1.0 / 3.0)
    cells = {}
    pts = np.empty((n, 3), dtype=np.float64)
    m = 0
    inv = 1.0 / min_sep
    while m < n:
        cand = rng.uniform(0.0, side, size=(n, 3))
        for x in cand:
            c = (int(x[0] * inv), int(x[1] * inv), int(x[2] * inv))
            ok = True
            for dx in (-1, 0, 1):
                for dy in (-1, 0, 1):
                    for dz in (-1, 0, 1):
                        for j in cells.get((c[0] + dx, c[1] + dy, c[2] + dz), ()):
                            d = pts[j] - x
                            if d[0] * d[0] + d[1] * d[1] + d[2] * d[2] < min_sep * min_sep:
                                ok = False
            if ok:
                pts[m] = x
                cells.setdefault(c, []).append(m)
                m += 1
                if m == n:
                    break
    return pts.astype(np.float32)


def morton_order(X, bits=10):
    """Permutation sorting points along a Z-order (Morton) curve: consecutive atoms become spatial neighbours,
    as consecutive atoms of a real chain are."""
    Xn = np.asarray(X, dtype=np.float64)
    lo, hi = Xn.min(0), Xn.max(0)
    cells = np.minimum(((Xn - lo) / np.maximum(hi - lo, 1e-9) * (1 << bits)).astype(np.int64), (1 << bits) - 1)
    code = np.zeros(Xn.shape[0], dtype=np.int64)
    for b in range(bits):
        for c in range(3):
            code |= ((cells[:, c] >> b) & 1) << (3 * b + c)
    return np.argsort(code, kind="stable")


def synthetic_structure(n, seed=1, n0=30, atoms_per_res=8, k=64, order="random", topology=True):
    """One synthetic structure with the reference's per-structure contract (before collation):
    X float32 [n,3]; ids_topk int64 [n, min(k,n)] 0-based (None with topology=False: the caller builds it, e.g. on the GPU with
    Model.knn_collate); q float32 one-hot [n, n0]; M bool [n, R].
    order="random": atoms in generation order (no spatial locality; what the golden fixtures use);
    order="morton": the same cloud with atoms renumbered along a Z-order curve (chain-like locality)."""
    X = synthetic_cloud(n, seed)
    if order == "morton":
        X = np.ascontiguousarray(X[morton_order(X)])
    elif order != "random":
        raise ValueError("order must be 'random' or 'morton'")
    rng = np.random.default_rng(seed + 7919)
    el = rng.choice(30, size=n, p=ELEMENT_P)
    q = np.zeros((n, n0), dtype=np.float32)
    q[np.arange(n), el] = 1.0
    if n0 == 123:  # resname (29+1) and atom-name (63+1) one-hots
        q[np.arange(n), 30 + rng.integers(0, 29, n)] = 1.0
        q[np.arange(n), 59 + rng.integers(0, 64, n)] = 1.0
    resid = np.arange(n) // atoms_per_res
    M = resid[:, None] == np.unique(resid)[None, :]
    return X, (extract_topology(X, k) if topology else None), q, M
