"""Multi-rank path on CPU: world_size-2 gloo process group, structures sharded by pesto_amd.sharding, forward supplied by
the CPU oracle (the HIP forward needs a GPU; the sharding/gather logic is backend-independent). Results must equal the
single-process results BITWISE for the same per-structure batching (SURVEY 8e acceptance)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, weights
from pesto_amd import sharding
from pesto_amd.config import CONFIGS


def test_partition_is_balanced_and_complete():
    costs = [3000, 100, 2500, 2400, 800, 799, 64, 3052, 1641]
    parts = sharding.partition(costs, 4)
    assert sorted(i for p in parts for i in p) == list(range(len(costs)))
    loads = [sum(costs[i] for i in p) for p in parts]
    assert max(loads) - min(loads) <= max(costs)
    assert sharding.partition(costs, 4) == parts                      # deterministic
    assert sharding.partition([5, 5], 4)[2:] == [[], []]              # more ranks than work


def test_batches_respect_atom_budget():
    sizes = [300, 200, 600, 100, 50]
    assert sharding.batches([0, 1, 2, 3, 4], sizes, 600, pack=False) == [[0, 1], [2], [3, 4]]      # input order
    b = sharding.batches([0, 1, 2, 3, 4], sizes, 600)                 # first-fit decreasing: fuller launches
    assert b == [[0, 1, 3], [2], [4]]
    assert all(sum(sizes[i] for i in g) <= 600 for g in b) and sorted(i for g in b for i in g) == [0, 1, 2, 3, 4]
    assert sharding.batches([2], sizes, 10) == [[2]]                  # an oversize structure still runs alone
    rng = np.random.default_rng(0)
    sz = [int(v) for v in rng.integers(1641, 3053, 53)]
    packed, seq = sharding.batches(list(range(53)), sz, 24576), sharding.batches(list(range(53)), sz, 24576, pack=False)
    assert len(packed) <= len(seq) and all(sum(sz[i] for i in g) <= 24576 for g in packed)


def _structures():
    from pesto_amd.topology import synthetic_structure
    return [synthetic_structure(n, seed) for n, seed in ((70, 1), (96, 2), (65, 3), (120, 4), (80, 5))]


def _oracle_forward():
    from oracle import oracle
    from pesto_amd.topology import mask_to_segments
    m = oracle.OracleModel(CONFIGS["i_v4_0"], weights("i_v4_0"))

    def fwd(X, ids, q, M):
        roa, R = mask_to_segments(M)
        return m.forward_segments(X, np.asarray(ids).astype(np.int32), q, roa, R, stop_after=3)
    return fwd


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["OMP_NUM_THREADS"] = "2"
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        res = sharding.forward_sharded(_oracle_forward(), _structures(), n_out=5, max_atoms=200)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **{str(i): z for i, z in enumerate(res)})
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_equals_single_process(tmp_path):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    # single-process reference with the SAME per-structure batching the ranks used
    structures = _structures()
    fwd = _oracle_forward()
    sizes = [s[0].shape[0] for s in structures]
    single = {}
    for part in sharding.partition(sizes, 2):
        single.update(sharding.forward_local(fwd, structures, part, max_atoms=200))
    for rank in range(2):
        got = np.load(os.path.join(str(tmp_path), f"rank{rank}.npz"))
        assert len(got.files) == len(structures)
        for i in range(len(structures)):
            assert np.array_equal(got[str(i)], single[i]), (rank, i)   # bitwise
            assert got[str(i)].shape == (structures[i][3].shape[1], 5)


def _bcast_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from pesto_amd import Model
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        m = Model(CONFIGS["i_v4_0"])                                  # (no handle is built: there is no GPU here)
        info = sharding.broadcast_weights(m, weights("i_v4_0") if rank == 0 else None, src=0)      # only rank 0 holds a checkpoint
        ranks = sharding.describe_ranks()
        np.savez(os.path.join(out_dir, f"b{rank}.npz"), blob=m.blob(), sha=np.array(info["sha256_16"]), seen=ranks["ranks_seen"],
                 world=ranks["world"], equal=info["ranks_equal"], order=np.array([d["rank"] for d in ranks["devices"]]))
    finally:
        dist.destroy_process_group()


def test_weights_are_broadcast_from_rank0_and_checked(tmp_path):
    """SURVEY 8e "init: ncclBroadcast(weights)": rank 0 alone has the state_dict; every rank ends with the same blob (checksum all-gathered),
    which is the blob load_state_dict builds; describe_ranks counts the ranks through an all_reduce (gloo here, RCCL on the GPU box)."""
    import torch.multiprocessing as mp
    from pesto_amd.weights import flatten_state_dict
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_bcast_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    want = np.ascontiguousarray(flatten_state_dict(CONFIGS["i_v4_0"], weights("i_v4_0")), dtype=np.float32)
    for rank in range(2):
        g = np.load(os.path.join(str(tmp_path), f"b{rank}.npz"))
        assert np.array_equal(g["blob"], want) and int(g["seen"]) == 2 and int(g["world"]) == 2 and bool(g["equal"])
        assert list(g["order"]) == [0, 1]
    assert str(np.load(os.path.join(str(tmp_path), "b0.npz"))["sha"]) == str(np.load(os.path.join(str(tmp_path), "b1.npz"))["sha"])
    # without a process group: plain load_state_dict, same blob
    from pesto_amd import Model
    m = Model(CONFIGS["i_v4_0"])
    info = sharding.broadcast_weights(m, weights("i_v4_0"))
    assert np.array_equal(m.blob(), want) and info["backend"] is None
    with pytest.raises(ValueError):
        Model(CONFIGS["i_v4_0"]).load_blob(want[:-1])


def test_padded_structures_get_launches_of_their_own():
    """precision "auto" repeats a structure with zero-padded neighbour slots on the exact kernels, and that repeat covers its whole launch:
    forward_local keeps such structures (fewer than 64 atoms, or a table of fewer than 64 columns) out of the launches of the others."""
    from pesto_amd.topology import synthetic_structure
    structures = [synthetic_structure(n, 7 + i) for i, n in enumerate((120, 30, 200, 64, 17, 90))]
    structures.append((structures[0][0], structures[0][1][:, :8], structures[0][2], structures[0][3]))      # 120 atoms, 8 columns
    seen = []

    class Recorder:
        def forward_batch(self, structs, independent=True):
            seen.append([s[0].shape[0] if np.shape(s[1])[1] == 64 or s[0].shape[0] < 64 else -s[0].shape[0] for s in structs])
            return [np.zeros((np.asarray(s[3]).shape[1], 5), np.float32) for s in structs]
    res = sharding.forward_local(Recorder(), structures, list(range(len(structures))), max_atoms=10000)
    assert len(res) == len(structures)
    assert sorted(map(sorted, seen)) == sorted([sorted([120, 200, 64, 90]), sorted([30, 17, -120])])


def test_failed_structure_is_skipped_not_fatal():
    structures = _structures()[:3]
    calls = {"n": 0}

    def flaky(X, ids, q, M):
        calls["n"] += 1
        if np.asarray(X).shape[0] in (96, 70 + 96):      # batch containing structure 1, and structure 1 alone
            raise ValueError("boom")
        return np.zeros((np.asarray(M).shape[1], 5), np.float32)
    res = sharding.forward_local(flaky, structures, [0, 1, 2], max_atoms=170)
    assert res[1] is None and res[0] is not None and res[2] is not None


def test_systemic_failure_propagates():
    """Only per-structure input errors (ValueError, PestoError with PESTO_ERR_INVALID / PESTO_ERR_RANGE) are skipped; anything else
    - a PestoError carrying PESTO_ERR_HIP / PESTO_ERR_NOMEM or no code, any other exception - or a rank on which EVERY structure
    fails, is an error of the run, not of a structure."""
    from pesto_amd._lib import PestoError
    structures = _structures()[:3]

    def make(code):
        def fwd(X, ids, q, M):
            if np.asarray(X).shape[0] in (96, 70 + 96):      # the batch containing structure 1, and structure 1 alone
                e = PestoError("boom")
                e.code = code
                raise e
            return np.zeros((np.asarray(M).shape[1], 5), np.float32)
        return fwd
    for code in (-1, -5):                                  # bad input / out of the f16 range: that structure only
        res = sharding.forward_local(make(code), structures, [0, 1, 2], max_atoms=170)
        assert res[1] is None and res[0] is not None and res[2] is not None
    for code in (-2, -3, None):                            # HIP failure, out of memory, library missing: the run
        with pytest.raises(PestoError):
            sharding.forward_local(make(code), structures, [0, 1, 2], max_atoms=170)
    structures = structures[:2]

    def broken(X, ids, q, M):
        raise OSError("library missing")
    with pytest.raises(OSError):
        sharding.forward_local(broken, structures, [0, 1], max_atoms=100)

    def all_bad(X, ids, q, M):
        raise ValueError("bad input")
    with pytest.raises(ValueError):
        sharding.forward_local(all_bad, structures, [0, 1], max_atoms=100)


def _all_bad_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import datetime

    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    structures = _structures()[:2]                       # one structure per rank (LPT): rank 1 owns the 70-atom one

    def fwd(X, ids, q, M):
        if np.asarray(X).shape[0] == 70:
            raise ValueError("bad input")
        return np.zeros((np.asarray(M).shape[1], 5), np.float32)
    try:
        try:
            sharding.forward_sharded(fwd, structures, n_out=5, max_atoms=200)
            verdict = "returned"
        except sharding.AllStructuresFailed as e:
            verdict = "raised:" + str(e)
        open(os.path.join(out_dir, f"verdict{rank}.txt"), "w").write(verdict)
    finally:
        dist.destroy_process_group()


def test_rank_whose_structures_all_fail_does_not_hang_the_collective(tmp_path):
    """A rank that owns only bad structures still enters the gather; afterwards EVERY rank raises the same error (before this the
    failing rank raised in front of the collective and the others waited for the process-group timeout)."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_all_bad_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    v = [open(os.path.join(str(tmp_path), f"verdict{r}.txt")).read() for r in range(2)]
    assert all(x.startswith("raised:") for x in v) and v[0] == v[1]


# ---------------------------------------------------------------------------------------------- on the GPU box
def _hip_structures():
    from pesto_amd.topology import extract_topology, synthetic_structure
    out = []
    for i, n in enumerate((700, 90, 1300, 40, 260, 1100, 64, 500)):
        X, _, q, M = synthetic_structure(n, 300 + i)
        out.append((X, extract_topology(X, 64), q, M))
    return out


def _hip_model():
    from pesto_amd import Model
    m = Model(CONFIGS["i_v4_0"])
    m.load_state_dict(weights("i_v4_0"))
    return m


def _hip_worker(rank, world, port, out_dir, backend):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    # gloo on a 1-GPU box: every rank drives GPU 0 (gloo carries the collectives); nccl (= RCCL): one GPU per rank
    gpu = rank if backend == "nccl" else 0
    torch.cuda.set_device(gpu)
    kw = {"device_id": torch.device("cuda", gpu)} if backend == "nccl" else {}
    dist.init_process_group(backend, init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, **kw)
    try:
        m = _hip_model().to(f"cuda:{gpu}")
        res = sharding.forward_sharded(m, _hip_structures(), n_out=5, max_atoms=2000)
        # (default policy: the member of 40 atoms is repeated on the exact kernels by the pad trigger - on whichever rank owns it, and in
        # the one-call-per-structure run it is compared with: per-structure guard words keep the grouping-independence bitwise)
        assert m.status()["n_fp32_rerun"] <= 1
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **{str(i): z for i, z in enumerate(res)})
    finally:
        dist.destroy_process_group()


def _n_gpus():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


@pytest.mark.gpu
@pytest.mark.parametrize("backend,world", [("gloo", 2), ("nccl", 1), ("nccl", "all")])
def test_sharded_hip_forward_equals_single_process_bitwise(tmp_path, backend, world):
    """The REAL path under torch.distributed: ranks shard the structures, run them through libpesto_hip.so and gather every
    result. gloo / world 2 (both ranks on the box's one GPU) exercises the partition + ragged gather across processes; nccl /
    world 1 exercises the RCCL collectives with device tensors (what an 8-GPU node uses). Either way every structure must come
    back with exactly the bits of a plain single-process run, including the N < 64 members (PESTO_BATCH_INDEPENDENT).
    ("nccl", "all"): one rank per visible GPU over RCCL / xGMI - the real multi-GPU leg; skipped on a box with fewer than two."""
    import torch.multiprocessing as mp
    if world == "all":
        world = _n_gpus()
        if world < 2:
            pytest.skip("needs at least two GPUs (the round-end multi-GPU box runs it)")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_hip_worker, args=(world, port, str(tmp_path), backend), nprocs=world, join=True)
    structures = _hip_structures()
    m = _hip_model()
    single = [m.forward_batch([st])[0] for st in structures]         # one call per structure
    for rank in range(world):
        got = np.load(os.path.join(str(tmp_path), f"rank{rank}.npz"))
        assert len(got.files) == len(structures)
        for i in range(len(structures)):
            assert np.array_equal(got[str(i)], single[i]), (rank, i)
    if backend == "nccl" and world >= 2:
        # the first multi-GPU box that runs `pytest -m gpu` produces SURVEY 8e's acceptance evidence by itself: the
        # driver-style launch of the fixed-list leg - RCCL saw every rank, every rank its own device, the sharded result bitwise equal to
        # the world-1 run of the same list on rank 0
        import json
        import subprocess
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--mode", "strong", "--steps", "3", "--warmup", "1", "--cpu-budget", "0",
               "--no-latency", "--config4-structures", "16", "--strong-structures", str(32 * world)]
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
        assert p.returncode == 0, p.stderr[-2000:]
        out = json.loads([l for l in p.stdout.splitlines() if l.strip()][-1])
        assert out["n_gpus"] == world and out["rccl_ranks_seen"] == world and out["ranks"]["distinct_devices"] == world and out["ranks"]["backend"] == "nccl"
        assert out["config4_sharded"]["bitwise_equal_to_world1"] is True and out["config4_strong"]["bitwise_equal_to_world1"] is True
        assert out["config4_strong"]["speedup_vs_world1"] > 1.0 and out["weights_broadcast"]["ranks_equal"] is True
        assert all("numa_node" in d and "cpus_allowed" in d for d in out["ranks"]["devices"])


def test_numa_binding_reads_sysfs_and_never_fails(tmp_path):
    """a rank is pinned to the CPUs of its GPU's NUMA node (sharding.bind_rank_to_numa): the node from
    /sys/bus/pci/devices/<bdf>/numa_node, the CPUs from the node's cpulist intersected with what the process may use; where sysfs gives
    nothing the call binds nothing and says why. Run against a fake sysfs tree; the process's affinity is restored afterwards."""
    from pesto_amd import sharding
    if not hasattr(os, "sched_getaffinity"):
        pytest.skip("no sched_getaffinity on this platform")
    assert sharding._parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11] and sharding._parse_cpulist("") == []
    before = os.sched_getaffinity(0)
    omp_before = os.environ.get("OMP_NUM_THREADS")
    mine = sorted(before)
    half = mine[:max(1, len(mine) // 2)]
    root = tmp_path / "sys"
    (root / "bus" / "pci" / "devices" / "0000:0a:00.0").mkdir(parents=True)
    (root / "bus" / "pci" / "devices" / "0000:0a:00.0" / "numa_node").write_text("1\n")
    (root / "bus" / "pci" / "devices" / "0000:0b:00.0").mkdir(parents=True)
    (root / "bus" / "pci" / "devices" / "0000:0b:00.0" / "numa_node").write_text("-1\n")
    (root / "devices" / "system" / "node" / "node1").mkdir(parents=True)
    (root / "devices" / "system" / "node" / "node1" / "cpulist").write_text(",".join(str(c) for c in half) + ",100000\n")      # (a CPU the process may not use)
    try:
        assert sharding.numa_node_of_pci("0000:0a:00", str(root)) == 1 and sharding.numa_node_of_pci("0000:0B:00.0", str(root)) is None
        assert sharding.numa_node_of_pci("0000:ff:00", str(root)) is None
        info = sharding._bind_to_node_of("0000:0a:00", {"numa_node": None, "cpus_allowed": len(before), "cpu_list_head": None, "bound": False, "why": None},
                                         str(root), True, None)
        assert info["bound"] is True and info["numa_node"] == 1 and info["cpus_allowed"] == len(half) and info["cpu_list_head"] == half[:4]
        assert os.sched_getaffinity(0) == set(half) and os.environ["OMP_NUM_THREADS"] == str(len(half))
        none = sharding._bind_to_node_of("0000:0b:00", {"numa_node": None, "cpus_allowed": None, "cpu_list_head": None, "bound": False, "why": None}, str(root))
        assert none["bound"] is False and "numa_node" in none["why"] and os.sched_getaffinity(0) == set(half)
        # the public entry point on a machine without a GPU: nothing bound, a reason given, no exception
        pub = sharding.bind_rank_to_numa()
        assert pub["bound"] in (False, True) and (pub["bound"] or pub["why"])
    finally:
        os.sched_setaffinity(0, before)
        if omp_before is None:
            os.environ.pop("OMP_NUM_THREADS", None)
        else:
            os.environ["OMP_NUM_THREADS"] = omp_before


@pytest.mark.gpu
def test_bench_launcher_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2` as the driver calls it (self-launch under torch.distributed.run), both ranks on GPU 0 with gloo: ONE JSON
    line on stdout, whole-job value, per-rank values, and the config-4 leg (8 structures per rank) bitwise equal to a world-1 run."""
    import json
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--same-gpu", "--backend", "gloo", "--steps", "2", "--warmup", "1",
           "--cpu-budget", "0", "--no-latency", "--config4-structures", "8", "--strong-structures", "24"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert len(out["per_rank"]) == 2 and out["rank_time_max_over_min"] >= 1.0
    c4 = out["config4_sharded"]
    assert c4["structures"] == 16 and c4["bitwise_equal_to_world1"] is True and c4["parity_max_abs_vs_reference"] < 1e-4
    assert len(c4["per_rank"]) == 2 and sum(r["structures"] for r in c4["per_rank"]) == 16
    assert "cpu_baseline" not in out          # (rank 0 at N = 1 only)
    # round 5: the line explains itself - what the collective library saw, where the weights came from, the reference-native input forms
    # and the FIXED-list leg SURVEY 8e's acceptance is about
    assert out["rccl_ranks_seen"] == 2 and out["ranks"]["world"] == 2 and out["ranks"]["backend"] == "gloo"
    assert [d["rank"] for d in out["ranks"]["devices"]] == [0, 1] and all("device_name" in d and "pci_bus_id" in d for d in out["ranks"]["devices"])
    # round 6: every rank reports the NUMA node of its GPU and the CPUs it is pinned to (None / "why" where sysfs gives no node)
    assert all("numa_node" in d and "cpus_allowed" in d and "numa_bound" in d for d in out["ranks"]["devices"])
    assert out["config"]["call"] == "Model.forward(X, ids_topk, q, M)" and out["segment_call"]["bitwise_equal_to_the_headline_call"] is True
    wb = out["weights_broadcast"]
    assert wb["ranks_equal"] is True and wb["backend"] == "gloo" and wb["bytes"] > 5_000_000 and len(wb["sha256_16"]) == 16
    assert "pre-reduced OUTSIDE" in c4["inputs"]
    dn = c4["dense_forms"]
    assert dn["structures"] == 16 and dn["bitwise_equal_to_world1"] is True and dn["parity_max_abs_vs_reference"] < 1e-4 and "INSIDE the timed region" in dn["inputs"]
    cs = out["config4_strong"]
    assert cs["structures"] == 24 and cs["bitwise_equal_to_world1"] is True and cs["speedup_vs_world1"] > 0 and cs["scaling"].startswith("strong")


def test_bench_prints_one_json_error_line_instead_of_a_traceback():
    """A failure anywhere before the result line (here: no GPU visible to the process) must still give the driver ONE parseable JSON line
    on stdout - with "error" and the stage that failed - and a non-zero exit code; the traceback goes to stderr."""
    import json
    import subprocess
    env = dict(os.environ)
    env["HIP_VISIBLE_DEVICES"] = ""; env["CUDA_VISIBLE_DEVICES"] = ""; env["ROCR_VISIBLE_DEVICES"] = ""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--cpu-budget", "0"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert p.returncode != 0 and len(lines) == 1, (p.returncode, p.stdout[-500:])
    out = json.loads(lines[0])
    assert out["value"] is None and "error" in out and out["error_stage"] == "device check" and out["n_gpus"] == 1
