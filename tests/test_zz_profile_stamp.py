"""The committed profile is of the committed kernels (runs last in the CPU suite: the file name sorts behind every other test).

profiles/traffic_i_v4_1_n3000_b8.json is what bench.py quotes `roofline.frac_rocprof`, `roofline.traffic` and the per-nn rocprof
fractions from; it is written by profiles/prof_round.sh (rocprofv3 kernel trace + separate --pmc passes in ONE gpurun call) and stamped
with the sha256 of the kernel sources. A kernel edit without a fresh profile fails HERE instead of silently printing nulls in the
bench line (round 4 shipped two kernel commits that way)."""
import json
import os

import bench
from conftest import ROOT


def _stamp():
    return json.load(open(os.path.join(ROOT, "profiles", "traffic_i_v4_1_n3000_b8.json")))


def test_traffic_file_is_stamped():
    t = _stamp()
    assert len(t["source_hash"]) == 16 and any("k_edge<64" in k for k in t["kernel_symbols"])
    per = [v for k, v in t["kernels"].items() if "k_edge<64" in k][0]
    assert per["dispatches_per_forward"] == 8 and per["write_bytes_per_dispatch"] > 0
    assert t["source_hash"] == bench.source_hash(), (
        "profiles/traffic_i_v4_1_n3000_b8.json was collected on other kernel sources: run `gpurun -- bash profiles/prof_round.sh <tag>` "
        "and `bash profiles/install_round.sh <tag>` on this tree, commit the files")


def test_stamp_carries_the_kernel_trace_of_every_layer_kernel():
    t = _stamp()
    tr = t["rocprof_kernel_trace"]["kernels"]
    for nn in (8, 16, 32, 64):
        hit = [v for k, v in tr.items() if f"k_edge<{nn}," in k]
        assert hit and hit[0]["calls"] >= 8 and hit[0]["avg_ns"] > 0, nn
        assert any(f"k_edge<{nn}," in k for k in t["kernels"]), nn


def test_stamp_carries_the_issue_floor_the_clock_and_the_exact_kernels():
    """roofline.issue_floor_ratio / clock_GHz (per layer kernel) and the exact fp32 kernels' trace come from the same
    stamped file, so they are non-null in the bench line exactly when the stamp matches the tree."""
    t = _stamp()
    per = t["issue_floor"]["per_nn"]
    for nn in ("8", "16", "32", "64"):
        assert 1.5 < per[nn]["clock_GHz"] < 3.0 and per[nn]["predicted_us"] > 0 and per[nn]["measured_over_predicted"] > 0.8, nn
    fp32 = t["rocprof_kernel_trace_fp32"]["kernels"]
    assert any("k_edge<64,4,false" in k for k in fp32) and all(v["avg_ns"] > 0 for v in fp32.values())
    table, _ = bench.per_nn_table({"sum": [{"nn": 64}]}, {"edge_nn64": {"launches_per_forward": 8, "avg_launch_ms": 0.25}}, 24001, t)
    row = table["64"]
    assert row["hbm_actual_frac"] is not None and row["issue_floor_ratio"] is not None and row["clock_GHz"] is not None
