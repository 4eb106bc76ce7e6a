"""Input-contract helpers (pesto_amd/topology.py) against the reference's extract_topology /
collate_batch_features outputs stored in the golden fixtures. No GPU needed."""
import numpy as np
import pytest

from conftest import golden
from pesto_amd import topology as T


def test_extract_topology_matches_reference():
    g = golden("topology_synth300")
    ids = T.extract_topology(g["X"], 64)
    assert ids.shape == (300, 64) and ids.dtype == np.int64
    assert np.array_equal(ids, g["ids_topk0"])
    assert not np.any(ids == np.arange(300)[:, None])          # self excluded when N > 64


def test_extract_topology_small_n():
    g = golden("topology_synth50")
    ids = T.extract_topology(g["X"], 64)
    assert ids.shape == (50, 50)                                # knn = min(64, N)
    # self sorts to the far end, tied with the farthest atom: compare everything but the last two columns
    assert np.array_equal(ids[:, :48], g["ids_topk0"][:, :48])
    assert np.array_equal(np.sort(ids, 1), np.sort(g["ids_topk0"], 1))


def test_tree_path_equals_dense_path():
    X = T.synthetic_cloud(700, 5)
    assert np.array_equal(T._topology_tree(X, 64), T._topology_dense(X, 64))
    X[17] = X[400]                                               # coincident pair -> masked to the far end
    a, b = T._topology_tree(X, 64), T._topology_dense(X, 64)
    assert 400 not in a[17] and 17 not in a[400] and 17 not in a[17]
    # atoms 17 and 400 are now exactly tied as neighbours of everyone else: compare distances, not ids
    da = np.linalg.norm(X[a] - X[:, None], axis=2)
    db = np.linalg.norm(X[b] - X[:, None], axis=2)
    assert np.array_equal(da, db)


def test_collate_contract_matches_reference_fixture():
    g = golden("edge_batch2")
    (n0, r0), (n1, r1) = g["sizes"]
    X = g["X"]
    a = [X[:n0], T.extract_topology(X[:n0], 64), np.zeros((n0, 30), np.float32), np.ones((n0, r0), bool)]
    b = [X[n0:], T.extract_topology(X[n0:], 64), np.zeros((n1, 30), np.float32), np.ones((n1, r1), bool)]
    Xc, ids, q, M = T.collate_batch_features([a, b])
    assert np.array_equal(ids[:n0], g["ids_topk"][:n0])          # 1-based, offset
    # N=40 structure: self ties with the farthest atom at the far end (last two of its 40 columns)
    assert np.array_equal(ids[n0:, :n1 - 2], g["ids_topk"][n0:, :n1 - 2])
    assert np.array_equal(np.sort(ids[n0:], 1), np.sort(g["ids_topk"][n0:], 1))
    assert ids.shape == (n0 + n1, 64) and np.all(ids[n0:, n1:] == 0)
    assert M.shape == (n0 + n1, r0 + r1) and M[:n0, r0:].sum() == 0 and M[n0:, :r0].sum() == 0
    assert np.array_equal(Xc, X)


def test_mask_to_segments_and_validation():
    g = golden("edge_single_atom_residue")
    roa = g["res_of_atom"]
    R = int(roa.max()) + 1
    M = np.zeros((roa.size, R), np.float32)
    M[np.arange(roa.size), roa] = 1
    r2, R2 = T.mask_to_segments(M)
    assert R2 == R and np.array_equal(r2, roa)
    bad = M.copy(); bad[3] = 0
    with pytest.raises(ValueError):
        T.mask_to_segments(bad)
    bad = np.concatenate([M, np.zeros((roa.size, 1), np.float32)], 1)
    with pytest.raises(ValueError):
        T.mask_to_segments(bad)


def test_synthetic_structure_contract():
    X, ids, q, M = T.synthetic_structure(200, seed=4)
    assert X.dtype == np.float32 and X.shape == (200, 3)
    assert ids.shape == (200, 64) and q.shape == (200, 30) and np.all(q.sum(1) == 1)
    assert M.shape == (200, 25) and np.all(M.sum(1) == 1)
    d = np.linalg.norm(X[:, None] - X[None], axis=2) + np.eye(200) * 10
    assert d.min() >= 0.5
    X2 = T.synthetic_structure(200, seed=4)[0]
    assert np.array_equal(X, X2)
    q123 = T.synthetic_structure(100, seed=4, n0=123)[2]
    assert q123.shape == (100, 123) and np.all(q123.sum(1) == 3)


def test_mask_to_segments_native_pass_equals_the_numpy_contract():
    """pesto_io_mask_to_segments (one native pass, used by mask_to_segments for host float32 masks) against the numpy statement of the
    same contract: one member (> 0.5) per row, no empty column."""
    import pytest
    from pesto_amd.topology import mask_to_segments
    rng = np.random.default_rng(5)
    for n, R in ((1, 1), (17, 5), (300, 40), (1000, 333)):
        roa = np.concatenate([np.arange(R), rng.integers(0, R, n - R)]) if n >= R else np.arange(n)
        roa = rng.permutation(roa)[:n]
        if len(set(roa.tolist())) < R:
            roa[:R] = np.arange(R)
        M = np.zeros((n, R), np.float32)
        M[np.arange(n), roa] = 1.0
        got, Rg = mask_to_segments(M)
        assert Rg == R and got.dtype == np.int32 and np.array_equal(got, roa)
        got64, _ = mask_to_segments(M.astype(np.float64))          # (the numpy path: other dtypes / layouts)
        assert np.array_equal(got64, roa)
        for Mb in (M > 0.5, (M > 0.5).astype(np.uint8)):            # the mask as encode_structure returns it (bool), and as bytes
            gotb, Rb = mask_to_segments(Mb)
            assert Rb == R and gotb.dtype == np.int32 and np.array_equal(gotb, roa)
        Ms = np.ascontiguousarray(M[np.argsort(roa, kind="stable")])      # contiguous residues (the reference's layout): the hinted fast path
        assert np.array_equal(mask_to_segments(Ms)[0], np.sort(roa)) and np.array_equal(mask_to_segments(Ms > 0.5)[0], np.sort(roa))
    M = np.zeros((6, 3), np.float32); M[np.arange(6), [0, 1, 2, 0, 1, 2]] = 1.0
    for bad in ("two", "none", "empty"):
        Mb = M.copy()
        if bad == "two":
            Mb[2, 0] = 1.0
        elif bad == "none":
            Mb[4] = 0.0
        else:
            Mb[[2, 5]] = 0.0; Mb[[2, 5], 0] = 1.0
        with pytest.raises(ValueError):
            mask_to_segments(Mb)
        with pytest.raises(ValueError):
            mask_to_segments(Mb.astype(np.float64))
        with pytest.raises(ValueError):
            mask_to_segments(Mb > 0.5)
    # float rows the integer word sums cannot vouch for take the element-wise contract (> 0.5): 0.75 is a member, -0.0 and 0.25 are not
    Mo = np.zeros((3, 3), np.float32); Mo[0, 0] = 0.75; Mo[1, 1] = 1.0; Mo[1, 2] = -0.0; Mo[2, 2] = 1.0; Mo[2, 0] = 0.25
    assert np.array_equal(mask_to_segments(Mo)[0], [0, 1, 2])
    Mo[2, 0] = 0.6
    with pytest.raises(ValueError):
        mask_to_segments(Mo)
