"""Pins of the randomised parity sweep (tests/golden/fuzz_pins.npz, written by tests/golden/make_fuzz_pins.py from the imported reference).

Round 3's sweep (profiles/fuzz_vs_oracle.py) saw |hip - oracle| = 5.2e-4 and 3.1e-4 on two inputs, above the north-star's 1e-4. The
generator replays that sweep, recovers both inputs and runs the REFERENCE on them in fp32 (1 and 8 threads) and fp64:
  (a) collated batch [1023, 2000, 2 atoms] (the 2-atom member: 62 of 64 neighbour slots wrap, src/model_operations.py:8):
      reference fp32 vs fp64 1.15e-4 (1 vs 8 threads: 5.9e-5), |z| up to 14.8, states |p| ~ 50 at layer 15;
  (b) 500 atoms with a neighbour table of k = 8 columns zero-padded to 64 (:230): reference fp32 vs fp64 9.2e-5.
Both are ill-conditioned for ANY fp32 evaluation (10x the 1e-5 spread of protein inputs). The round-3 deviation was the CHECKER's: the
oracle's sequential float32 sums are 2 - 3x noisier per layer than the reference's blocked kernels there (oracle vs fp64: 6.9e-4 /
2.5e-4); its wide build (double accumulators, float32 storage) is within 4.5e-5 / 2.1e-5 of the fp64 reference.
Bound used here, as the round-3 verdict set it: |x - reference_fp64| <= max(1e-4, 2 |reference_fp32 - reference_fp64|); the seeded
fuzz leg holds the plain 1e-4 wherever the reference's own spread is below 1e-5.
"""
import numpy as np
import pytest

from conftest import golden, onehot, weights
from pesto_amd.config import CONFIGS

TAG = "i_v4_0"


def bound(z32, z64):
    return max(1e-4, 2.0 * float(np.abs(z32 - z64).max()))


def pinned(pre):
    g = golden("fuzz_pins")
    roa = g[pre + "roa"].astype(np.int32)
    return dict(X=g[pre + "X"], ids=g[pre + "ids"].astype(np.int32), q0=onehot(g[pre + "q"][:, None], 30), roa=roa, R=int(roa.max()) + 1,
                z32=g[pre + "z32_t8"], z32_t1=g[pre + "z32_t1"], z64=g[pre + "z64"])


def leg_round(it):
    """One round of the seeded fuzz leg: the collated batch (the reference's collate_batch_features output) and its members in the
    per-structure contract (0-based ids with the structure's own k columns), with the reference's fp32 / fp64 logits for both forms."""
    g = golden("fuzz_pins")
    pre = f"leg{it}_"
    X, ids, q, roa = g[pre + "X"], g[pre + "ids"].astype(np.int32), g[pre + "q"], g[pre + "roa"].astype(np.int32)
    sizes, ks = g[pre + "sizes"], g[pre + "k"]
    structs, a0, r0 = [], 0, 0
    for (n, R), k in zip(sizes, ks):
        n, R, k = int(n), int(R), int(k)
        M = np.zeros((n, R), np.float32)
        M[np.arange(n), roa[a0:a0 + n] - r0] = 1.0
        structs.append((X[a0:a0 + n], np.ascontiguousarray(ids[a0:a0 + n, :k] - a0 - 1), onehot(q[a0:a0 + n, None], 30), M))
        a0 += n; r0 += R
    roffs = np.concatenate([[0], np.cumsum(sizes[:, 1])]).astype(int)
    return dict(X=X, ids=ids, q0=onehot(q[:, None], 30), roa=roa, R=int(roffs[-1]), structs=structs, roffs=roffs,
                col_z32=g[pre + "col_z32"], col_z64=g[pre + "col_z64"], ind_z32=g[pre + "ind_z32"], ind_z64=g[pre + "ind_z64"])


# ------------------------------------------------------------------ CPU: the oracle (both builds) against the reference's fp64 logits
@pytest.mark.parametrize("pre", ["a_", "b_"])
def test_oracle_on_the_two_pinned_inputs(pre):
    from oracle import oracle
    p = pinned(pre)
    wide = oracle.OracleModel(CONFIGS[TAG], weights(TAG), wide=True).forward_segments(p["X"], p["ids"], p["q0"], p["roa"], p["R"])
    assert np.abs(wide - p["z64"]).max() <= bound(p["z32"], p["z64"])
    assert np.abs(wide - p["z64"]).max() < 1e-4          # (measured 4.5e-5 / 2.1e-5: closer to fp64 than the reference's own fp32 run)
    # the float32-accumulating port is the noisy one on these inputs (6.9e-4 / 2.5e-4): that, not the kernels, was round 3's deviation
    plain = oracle.OracleModel(CONFIGS[TAG], weights(TAG)).forward_segments(p["X"], p["ids"], p["q0"], p["roa"], p["R"])
    assert np.abs(plain - p["z64"]).max() < 2e-3


@pytest.mark.parametrize("it", [0, 1, 2])
def test_oracle_on_the_seeded_fuzz_leg(it):
    from oracle import oracle
    L = leg_round(it)
    for wide in (False, True):
        o = oracle.OracleModel(CONFIGS[TAG], weights(TAG), wide=wide)
        z = o.forward_segments(L["X"], L["ids"], L["q0"], L["roa"], L["R"])
        for j in range(len(L["structs"])):
            sl = slice(L["roffs"][j], L["roffs"][j + 1])
            spread = float(np.abs(L["col_z32"][sl] - L["col_z64"][sl]).max())
            tol = 1e-4 if spread < 1e-5 else max(1e-4, 2 * spread)
            assert np.abs(z[sl] - L["col_z64"][sl]).max() <= tol, (it, j, wide)


# ------------------------------------------------------------------ GPU: the HIP path against the reference's fp64 logits
def _model(precision="auto", pad_trigger=False):
    """pad_trigger=False: the split kernels themselves under "auto" (no repeat of zero-padded structures) - what the accuracy / bit-identity
    assertions below are about; the policy has its own test."""
    from pesto_amd import Model
    m = Model(CONFIGS[TAG], precision=precision)
    m.load_state_dict(weights(TAG))
    return m.set_auto_pad_trigger(pad_trigger).eval()


@pytest.mark.gpu
@pytest.mark.parametrize("pre", ["a_", "b_"])
@pytest.mark.parametrize("precision", ["auto", "fp32"])
def test_hip_on_the_two_pinned_inputs(pre, precision):
    p = pinned(pre)
    m = _model(precision, pad_trigger=True)                                   # the DEFAULT policy
    z = m.forward_segments(p["X"], p["ids"], p["q0"], p["roa"], p["R"])      # the collated call, as the reference ran it
    err = float(np.abs(z - p["z64"]).max())
    print(f"\n   pinned {pre} {precision}: |hip - reference fp64| {err:.2e}, |hip - reference fp32| {np.abs(z - p['z32']).max():.2e}, "
          f"reference fp32 vs fp64 {np.abs(p['z32'] - p['z64']).max():.2e}, bound {bound(p['z32'], p['z64']):.2e}")
    # round 5: factor ONE on the reference's own fp32-vs-fp64 spread, and the deviation from the reference's fp32 run
    # itself (8 threads, what a drop-in caller compares with) is asserted, not printed: measured 8.3e-5 / 8.2e-5 and 1.9e-4 / 1.7e-4
    assert err <= max(1e-4, float(np.abs(p["z32"] - p["z64"]).max()))
    assert float(np.abs(z - p["z32"]).max()) <= 2e-4
    # both pins carry zero-padded neighbour slots (a: the 2-atom member of the collated call; b: a table of 8 columns) - that is where
    # ALL of their ill-conditioning sits (profiles/dev/pin_where.py: input a, split kernels: 1.3e-4 on the one residue of the 2-atom
    # member, <= 2e-5 on the other 206). "auto" repeats such calls on the exact kernels (pesto_set_auto_pad_trigger, on by default)
    assert m.status()["n_fp32_rerun"] == (1 if precision == "auto" else 0)
    if precision == "auto":
        assert err < 3e-5
        m2 = _model("auto", pad_trigger=False)                  # the split kernels themselves: within 2 x the reference's own fp32 spread
        z2 = m2.forward_segments(p["X"], p["ids"], p["q0"], p["roa"], p["R"])
        assert float(np.abs(z2 - p["z64"]).max()) <= bound(p["z32"], p["z64"]) and m2.status()["n_fp32_rerun"] == 0
        well_posed = np.ones(p["R"], bool)
        if pre == "a_":
            well_posed[-1] = False                              # every residue but the 2-atom member's
            assert float(np.abs(z2 - p["z64"])[well_posed].max()) < 3e-5


@pytest.mark.gpu
def test_auto_state_limit_repeats_exactly_the_structures_above_it():
    """pesto_set_auto_state_limit (conditioning trigger of "auto"): with the limit below a structure's state magnitude the structure is
    repeated on the exact fp32 kernels - its logits equal precision "fp32" bit for bit, the others keep the split kernels' bits; the
    default (128) fires on none of the pinned / fuzz inputs; <= 0 switches it off."""
    L = leg_round(0)
    m = _model()
    base = m.forward_batch(L["structs"], independent=True)
    assert m.status()["n_fp32_rerun"] == 0
    exact = _model("fp32").forward_batch(L["structs"], independent=True)
    m.set_auto_state_limit(32.0)                       # profiles/r05_state_limit.txt: one member of this round exceeds 32 (and 40), none 48
    z = m.forward_batch(L["structs"], independent=True)
    n = m.status()["n_fp32_rerun"]
    assert n == 1
    hit = [j for j in range(len(z)) if not np.array_equal(z[j], base[j])]
    assert len(hit) == 1 and np.array_equal(z[hit[0]], exact[hit[0]])
    m.set_auto_state_limit(0.0)
    assert all(np.array_equal(a, b) for a, b in zip(m.forward_batch(L["structs"], independent=True), base)) and m.status()["n_fp32_rerun"] == n
    m.set_precision("f16_split").set_auto_state_limit(1.0)          # f16_split never repeats, so it never flags: no range error either
    assert all(np.array_equal(a, b) for a, b in zip(m.forward_batch(L["structs"], independent=True), base))


@pytest.mark.gpu
@pytest.mark.parametrize("it", [0, 1, 2])
def test_hip_seeded_fuzz_leg(it):
    """Three seeded rounds of ragged batches (2 - 1,025 atoms incl. 63 / 64 / 65 / 66, neighbour tables of 3 - 64 columns, residues of
    1 - 30 atoms, permuted residue columns): COLLATED vs the reference's collated forward, INDEPENDENT vs the reference's one call per
    structure, both at 1e-4 where the reference's own fp32-vs-fp64 spread is below 1e-5 (else 2 x that spread); and INDEPENDENT ==
    one call per structure == the pipelined submit / wait path, bit for bit."""
    L = leg_round(it)
    m = _model()
    z_col = m.forward_segments(L["X"], L["ids"], L["q0"], L["roa"], L["R"])
    z_ind = m.forward_batch(L["structs"], independent=True)
    z_pipe = m.forward_batch_wait(m.forward_batch_submit(L["structs"], independent=True))
    z_colb = np.concatenate(m.forward_batch(L["structs"], independent=False), 0)
    assert np.array_equal(z_col, z_colb)                      # device collate == the reference's collate_batch_features
    worst = 0.0
    for j, st in enumerate(L["structs"]):
        sl = slice(L["roffs"][j], L["roffs"][j + 1])
        for name, z, z32, z64 in (("collated", z_col[sl], L["col_z32"][sl], L["col_z64"][sl]), ("independent", z_ind[j], L["ind_z32"][sl], L["ind_z64"][sl])):
            spread = float(np.abs(z32 - z64).max())
            tol = 1e-4 if spread < 1e-5 else max(1e-4, 2 * spread)
            err = float(np.abs(z - z64).max())
            worst = max(worst, err)
            assert err <= tol, (it, j, name, st[0].shape[0], err, spread)
        single = m.forward_batch([st], independent=True)[0]
        assert np.array_equal(z_ind[j], single) and np.array_equal(z_pipe[j], single), ("bitwise", it, j)
    print(f"\n   fuzz leg round {it}: max |hip - reference fp64| = {worst:.2e}")
    assert m.status()["n_fp32_rerun"] == 0


@pytest.mark.gpu
def test_auto_pad_trigger_repeats_exactly_the_padded_structures():
    """pesto_set_auto_pad_trigger (default on): under "auto" a structure with zero-padded neighbour slots - fewer than 64 atoms, or a
    table of fewer than 64 columns - is repeated on the exact fp32 kernels (equal to precision "fp32" bit for bit), the others keep the
    split kernels' bits; a COLLATED call is one structure for the guard: one padded member repeats the call. F16_SPLIT never flags."""
    L = leg_round(0)
    structs = L["structs"]
    padded = [st[0].shape[0] < 64 or st[1].shape[1] < 64 for st in structs]
    assert any(padded) and not all(padded)
    split = _model("auto", pad_trigger=False).forward_batch(structs, independent=True)
    exact = _model("fp32").forward_batch(structs, independent=True)
    m = _model("auto", pad_trigger=True)
    z = m.forward_batch(structs, independent=True)
    assert m.status()["n_fp32_rerun"] == sum(padded)
    for j, pd in enumerate(padded):
        assert np.array_equal(z[j], exact[j] if pd else split[j]), (j, pd)
        assert np.array_equal(z[j], m.forward_batch([structs[j]], independent=True)[0]), j         # grouping-independent, bit for bit
    n0 = m.status()["n_fp32_rerun"]
    zc = m.forward_segments(L["X"], L["ids"], L["q0"], L["roa"], L["R"])                            # collated: one guard word
    assert m.status()["n_fp32_rerun"] == n0 + 1
    assert np.array_equal(zc, _model("fp32").forward_segments(L["X"], L["ids"], L["q0"], L["roa"], L["R"]))
    mf = _model("f16_split", pad_trigger=True)
    assert all(np.array_equal(a, b) for a, b in zip(mf.forward_batch(structs, independent=True), split))


@pytest.mark.gpu
def test_pad_trigger_looks_at_the_columns_a_layer_reads():
    """the trigger compares against the model's LARGEST nn, not the table width of 64 - a k = 32 neighbour table under a model
    whose layers gather at most 16 neighbours has no padded slot any layer reads (no repeat, the split kernels' bits), the same structure
    with an 8-column table has (repeated on the exact kernels: equal to precision "fp32" bit for bit). Same bound in Model.max_nn, which
    the bulk loops' grouping (sharding.forward_local, apply) uses."""
    import copy
    from pesto_amd import Model
    from pesto_amd.topology import extract_topology, mask_to_segments, synthetic_structure
    from pesto_amd.weights import synthetic_state_dict
    cfg = copy.deepcopy(CONFIGS["i_v4_0"])
    cfg["sum"] = [dict(cfg["sum"][0], nn=8), dict(cfg["sum"][0], nn=16)]
    sd = synthetic_state_dict(cfg, seed=3)
    X, ids0, q, M = synthetic_structure(200, 9)
    roa, R = mask_to_segments(M)
    ids = (ids0 + 1).astype(np.int32)

    def model(precision):
        m = Model(cfg, precision=precision)
        m.load_state_dict(sd)
        return m.eval()

    assert model("auto").max_nn == 16
    for k, repeats in ((32, 0), (16, 0), (8, 1)):
        tab = np.ascontiguousarray(ids[:, :k])
        m = model("auto")
        z = m.forward_segments(X, tab, q, roa, R)
        assert m.status()["n_fp32_rerun"] == repeats, k
        ref = model("fp32" if repeats else "f16_split").forward_segments(X, tab, q, roa, R)
        assert np.array_equal(z, ref), k
