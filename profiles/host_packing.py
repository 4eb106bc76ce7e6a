#!/usr/bin/env python3
"""What the HOST side of the sharded bulk path costs, with the GPU idle (GPU box): python profiles/host_packing.py [processes] [seconds]

Eight ranks must pack 8 x 1,700 structures/s between them before the GPUs are the limit. Every process builds a handle, switches it
to pesto_debug_host_only (pesto_forward_batch_submit then does its host half only - argument checks, one-hot detection, packing into
the pinned staging slot - and queues nothing on the GPU) and pushes the config-4 work list (the 53 pdbs_test chains, compact forms as
bench.py's config-4 leg hands them over, launches of <= 24,576 atoms) through sharding.forward_local for a fixed time. Reported per
process and in total: structures/s of packing (wall clock), CPU seconds per structure (time.process_time of the process). A second pass
with the DENSE per-structure forms (float one-hot features, dense residue mask, int32 ids) shows what the library's in-pack one-hot
detection and the mask reduction cost (round 5: one native checked pass over the bool / float mask, ids narrowed by the packer).

    python profiles/host_packing.py numa [processes] [seconds]      (round 6)
the compact and the dense-native pass with the processes PLACED: unpinned, every process on NUMA node 0, every process on the last node,
processes spread round-robin over the nodes (what sharding.bind_rank_to_numa gives eight ranks whose GPUs hang off both sockets) - the
packing rate per placement, with the host's node / CPU map."""
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _node_cpus():
    """{node: [cpus]} of this host (sysfs), restricted to the CPUs this process may use"""
    from pesto_amd.sharding import _parse_cpulist
    import glob
    import re
    out = {}
    allowed = os.sched_getaffinity(0)
    for d in sorted(glob.glob("/sys/devices/system/node/node[0-9]*")):
        try:
            cpus = [c for c in _parse_cpulist(open(os.path.join(d, "cpulist")).read()) if c in allowed]
        except OSError:
            continue
        if cpus:
            out[int(re.search(r"node(\d+)$", d).group(1))] = cpus
    return out


def worker(rank, seconds, dense, q, cpus=None):
    if cpus:
        os.sched_setaffinity(0, cpus)
    import bench
    from pesto_amd import Model, sharding
    from pesto_amd.config import CONFIGS
    cfg = CONFIGS["i_v4_1"]
    sd, _ = bench.load_weights(cfg)
    m = Model(cfg, validate=False)
    m.load_state_dict(sd)
    # dense = "native": the reference loader's own per-structure outputs (float32 one-hot q, BOOL mask M, int64 ids:
    # src/data_encoding.py:61-102); "float": the same with M.float() and int32 ids (round 4's dense row)
    structures, sizes, _ = bench.config4_structures(64, m, "dense" if dense else "compact")           # (GPU k-NN once, outside the timed region)
    if dense == "float":
        structures = [(X, ids.astype(np.int32), q0, M.astype(np.float32)) for X, ids, q0, M in structures]
    if dense:
        m.validate = True
    m.debug_host_only(True)
    idx = list(range(len(structures)))
    sharding.forward_local(m, structures, idx)                       # warm-up: pinned slots allocated
    n, t0, c0 = 0, time.perf_counter(), time.process_time()
    while time.perf_counter() - t0 < seconds:
        sharding.forward_local(m, structures, idx)
        n += len(structures)
    q.put((rank, n, time.perf_counter() - t0, time.process_time() - c0))


def run(procs, seconds, dense, placement=None):
    """placement: None, or a list of CPU lists, one per process"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, seconds, dense, q, placement[r] if placement else None)) for r in range(procs)]
    for p in ps:
        p.start()
    res = sorted(q.get() for _ in ps)
    for p in ps:
        p.join()
    rate = [n / w for _, n, w, _ in res]
    forms = {False: "compact (uint8 feature indices, res_of_atom, uint16 ids)",
             "native": "dense, the reference loader's own outputs (float32 one-hot q, BOOL mask M, int64 ids; validate=True)",
             "float": "dense with M.float() and int32 ids (float32 one-hot q, float32 mask M; validate=True) - round 4's dense row"}[dense]
    return {"processes": procs, "forms": forms,
            "structures_per_s_total": float(sum(rate)), "structures_per_s_per_process": [float(r) for r in rate],
            "host_cpu_s_per_structure": float(np.mean([c / n for _, n, _, c in res])),
            "needed_for_8_gpus": 8 * 1700.0}


def numa_runs(procs, seconds):
    nodes = _node_cpus()
    out = {"host_cores": os.cpu_count(), "cpus_allowed": len(os.sched_getaffinity(0)), "numa_nodes": {str(k): {"cpus": len(v), "first": v[0], "last": v[-1]} for k, v in nodes.items()},
           "runs": []}
    ids = sorted(nodes)
    placements = {"unpinned": None}
    if ids:
        placements[f"all on node {ids[0]}"] = [nodes[ids[0]]] * procs
        if len(ids) > 1:
            placements[f"all on node {ids[-1]}"] = [nodes[ids[-1]]] * procs
            placements["spread round-robin over the nodes"] = [nodes[ids[r % len(ids)]] for r in range(procs)]
    for name, pl in placements.items():
        for dense in (False, "native"):
            r = run(procs, seconds, dense, pl)
            r["placement"] = name
            out["runs"].append(r)
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "numa":
        print(json.dumps(numa_runs(int(sys.argv[2]) if len(sys.argv) > 2 else 8, float(sys.argv[3]) if len(sys.argv) > 3 else 5.0), indent=1))
        sys.exit(0)
    procs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
    out = {"host_cores": os.cpu_count(), "runs": [run(1, seconds, False), run(procs, seconds, False), run(procs, seconds, "native"), run(procs, seconds, "float")]}
    print(json.dumps(out, indent=1))
