"""Structure-level sharding of a work list over the GPUs of one node.

Structures never interact in the forward pass (collation only concatenates and offsets indices, reference
src/dataset.py:102-110), so the path shards with NO data-path collective: every rank owns one MI355X, a full copy of
the weights and its own slice of the structures.  The reference's bulk driver is a plain Python loop over
structures (interfaceome/apply_model.py:57-82, apply_model.ipynb:139-167); this module is its multi-GPU form.

torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests) carries only
  * the ragged result gather: per-structure logits z_i [R_i, n_out] (a few KB each), as one all_gather of counts
    and one all_gather of padded payloads,
  * the weight broadcast at start-up (broadcast_weights: rank 0 reads the checkpoint, every rank builds its handle from the
    broadcast blob; SURVEY 8e "init: ncclBroadcast(weights)"), checked by a checksum all-gather,
  * barriers / the max-over-ranks timing reduction of bench.py.
"""
import logging

import numpy as np

log = logging.getLogger("pesto_amd.sharding")


def partition(costs, world_size):
    """Longest-processing-time-first assignment of work items to ranks. ``costs[i]`` ~ atoms of structure i
    (layer cost is O(N * nn)). Deterministic; returns a list of index lists, one per rank, each in ascending order."""
    costs = np.asarray(costs, dtype=np.float64)
    order = np.argsort(-costs, kind="stable")
    load = np.zeros(world_size)
    out = [[] for _ in range(world_size)]
    for i in order:
        r = int(np.argmin(load))
        out[r].append(int(i))
        load[r] += costs[i]
    return [sorted(ix) for ix in out]


def batches(indices, sizes, max_atoms, pack=True):
    """Grouping of a rank's structures into collated launches of at most ``max_atoms`` atoms (>= 1 structure each).
    pack=True: first-fit decreasing - launches as close to ``max_atoms`` as the sizes allow (the kernels work in rounds of 6,144
    atoms over the 256 CUs: the default 24,576 is four full rounds, and a launch of 20k atoms pays for four rounds as well), fewer
    launches than filling in input order. Results do not depend on the grouping (PESTO_BATCH_INDEPENDENT), so any packing is valid.
    pack=False: sequential filling in input order. Deterministic either way; groups are returned in order of their first index."""
    if not pack:
        out, cur, tot = [], [], 0
        for i in indices:
            if cur and tot + sizes[i] > max_atoms:
                out.append(cur)
                cur, tot = [], 0
            cur.append(i)
            tot += sizes[i]
        if cur:
            out.append(cur)
        return out
    bins, loads = [], []
    for i in sorted(indices, key=lambda j: (-sizes[j], j)):
        for b in range(len(bins)):
            if loads[b] + sizes[i] <= max_atoms:
                bins[b].append(i)
                loads[b] += sizes[i]
                break
        else:
            bins.append([i])
            loads.append(sizes[i])
    return sorted((sorted(b) for b in bins), key=lambda b: b[0])


# pesto_status codes that describe ONE structure's inputs (include/pesto_hip.h): PESTO_ERR_INVALID, PESTO_ERR_RANGE
_PER_STRUCTURE_CODES = (-1, -5)


def _is_skippable(exc):
    """Per-structure failures the bulk loop skips (the reference driver wraps each structure in try/except and continues,
    interfaceome/apply_model.py:57-82): bad inputs reported by the host checks (ValueError) or by the library with a per-structure
    status (PESTO_ERR_INVALID, PESTO_ERR_RANGE). Anything else - a missing library (no code), PESTO_ERR_HIP, PESTO_ERR_NOMEM - is
    systemic: retrying the remaining structures would silently truncate the results, so it propagates."""
    from ._lib import PestoError
    if isinstance(exc, PestoError):
        return exc.code in _PER_STRUCTURE_CODES
    return isinstance(exc, ValueError)


class AllStructuresFailed(RuntimeError):
    """Every structure of a rank failed its per-structure checks: a systemic problem, not a bad input."""


def forward_local(forward_fn, structures, indices, max_atoms=24576, raise_if_all_failed=True):
    """Run ``forward_fn(X, ids_topk, q, M) -> z`` (or a pesto_amd.Model) over this rank's structures, collating several per launch.
    ``structures[i] = (X, ids_topk0, q, M)`` with the per-structure contract of pesto_amd.topology.
    Returns {index: z_i (numpy [R_i, n_out])}.  With a Model the launches use PESTO_BATCH_INDEPENDENT: every structure gets
    the result of its own call whatever it was grouped with, so any partition over ranks gives the same bits.
    A structure whose batch raises a per-structure error (_is_skippable) is retried alone, logged and skipped (None) on a second
    failure; any other error propagates at once. If EVERY structure of this rank fails the last error is re-raised
    (raise_if_all_failed; forward_sharded defers that until after its collective so that no rank is left waiting in it)."""
    from .topology import collate_batch_features
    sizes = [np.asarray(s[0]).shape[0] for s in structures]
    results = {}

    def run(group):
        if hasattr(forward_fn, "forward_batch"):      # a pesto_amd.Model: collate on the device (pesto_forward_batch)
            for i, z in zip(group, forward_fn.forward_batch([tuple(structures[i]) for i in group], independent=True)):
                results[i] = np.ascontiguousarray(z)
            return
        X, ids, q, M = collate_batch_features([list(structures[i]) for i in group])
        z = forward_fn(X, ids, q, M)
        z = z.detach().cpu().numpy() if hasattr(z, "detach") else np.asarray(z)
        r0 = 0
        for i in group:
            r = np.asarray(structures[i][3]).shape[1]
            results[i] = np.ascontiguousarray(z[r0:r0 + r])
            r0 += r

    last_error = None
    # structures with zero-padded neighbour slots a layer reads (fewer atoms / table columns than the model's largest nn) get launches of their own: precision "auto" repeats such
    # structures on the exact fp32 kernels (pesto_set_auto_pad_trigger), and the repeat runs the exact kernels over the WHOLE launch of a
    # flagged structure (writing only its logits) - one peptide must not make 24,000 atoms of batch mates pay for it. Results do not
    # depend on the grouping (PESTO_BATCH_INDEPENDENT), so this is a cost decision only.
    pad_cols = int(getattr(forward_fn, "max_nn", 64))      # (a plain callable: the full table width)

    def _padded(i):
        ids = structures[i][1]
        return sizes[i] < pad_cols or (np.ndim(ids) == 2 and np.shape(ids)[1] < pad_cols)
    small = [i for i in indices if _padded(i)]
    groups = batches([i for i in indices if not _padded(i)], sizes, max_atoms) + (batches(small, sizes, max_atoms) if small else [])
    # a pesto_amd.Model: two launches in flight (submit t + 1 while t computes: host packing and the H2D copy overlap the kernels)
    pipelined = hasattr(forward_fn, "forward_batch_submit")
    pending = None      # (group, ticket) of the launch in flight

    def finish(p):
        g, t = p
        for i, z in zip(g, forward_fn.forward_batch_wait(t)):
            results[i] = np.ascontiguousarray(z)

    if pipelined:
        todo = []       # groups whose pipelined launch failed: rerun below on the synchronous path with its per-structure handling
        for group in groups:
            try:
                ticket = forward_fn.forward_batch_submit([tuple(structures[i]) for i in group], independent=True)
            except Exception as e:
                if not _is_skippable(e):
                    raise
                todo.append(group)
                continue
            if pending is not None:
                try:
                    finish(pending)
                except Exception as e:
                    if not _is_skippable(e):
                        raise
                    todo.append(pending[0])
            pending = (group, ticket)
        if pending is not None:
            try:
                finish(pending)
            except Exception as e:
                if not _is_skippable(e):
                    raise
                todo.append(pending[0])
        groups = todo
    for group in groups:
        try:
            run(group)
        except Exception as e_group:
            if not _is_skippable(e_group):
                raise
            if len(group) == 1:
                log.warning("structure %d skipped: %s", group[0], e_group)
                results[group[0]] = None
                last_error = e_group
                continue
            for i in group:
                try:
                    run([i])
                except Exception as e:
                    if not _is_skippable(e):
                        raise
                    log.warning("structure %d skipped: %s", i, e)
                    results[i] = None
                    last_error = e
    if raise_if_all_failed and indices and last_error is not None and all(results.get(i) is None for i in indices):
        raise last_error
    return results


def _collective_device(group=None):
    """Tensors of a collective must live where the backend works: RCCL ("nccl") needs device tensors, gloo host tensors."""
    import torch
    import torch.distributed as dist
    if dist.get_backend(group) == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def gather_results(local, n_total, n_out, group=None, device=None, failed_ranks=None):
    """All ranks receive every structure's z. ``local``: {index: array or None}. Three small collectives:
    all_gather of (index, rows) descriptors and row counts, all_gather of row payloads padded to the largest rank.
    ``device``: where the collective tensors live; default from the group's backend (cuda for nccl = RCCL, cpu for gloo).
    ``failed_ranks`` (optional list): receives the ranks that owned structures and delivered none (same answer on every rank)."""
    import torch
    import torch.distributed as dist
    if device is None:
        device = _collective_device(group)
    world = dist.get_world_size(group)
    items = sorted(local.items())
    desc = torch.full((n_total, 2), -1, dtype=torch.int64)
    rows = []
    for k, (i, z) in enumerate(items):
        desc[k, 0] = i
        desc[k, 1] = -1 if z is None else z.shape[0]
        if z is not None:
            rows.append(torch.from_numpy(np.asarray(z, dtype=np.float32)).reshape(-1, n_out))
    payload = torch.cat(rows, 0) if rows else torch.zeros((0, n_out))
    n_rows = torch.tensor([payload.shape[0]], dtype=torch.int64)
    desc, n_rows, payload = desc.to(device), n_rows.to(device), payload.to(device)
    all_desc = [torch.empty_like(desc) for _ in range(world)]
    all_rows = [torch.empty_like(n_rows) for _ in range(world)]
    dist.all_gather(all_desc, desc, group=group)
    dist.all_gather(all_rows, n_rows, group=group)
    max_rows = max(int(r.item()) for r in all_rows)
    padded = torch.zeros((max(max_rows, 1), n_out), dtype=torch.float32, device=device)
    padded[:payload.shape[0]] = payload
    all_pay = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(all_pay, padded, group=group)
    out = [None] * n_total
    for r in range(world):
        d = all_desc[r].cpu().numpy()
        pay = all_pay[r].cpu().numpy()
        off = 0
        owned = good = 0
        for i, nrow in d:
            if i < 0:
                break
            owned += 1
            if nrow >= 0:
                out[int(i)] = pay[off:off + nrow].copy()
                off += int(nrow)
                good += 1
        if failed_ranks is not None and owned > 0 and good == 0:
            failed_ranks.append(r)
    return out


def forward_sharded(forward_fn, structures, n_out, max_atoms=24576, group=None, device=None, timings=None):
    """Shard ``structures`` over the ranks of the (already initialised) process group, run them, gather all results
    on every rank.  Single-process (no process group): runs everything locally.
    ``timings`` (optional dict): receives this rank's seconds in its own launches ("local_s"), in the result gather ("gather_s") and
    the structures / atoms it owned - a straggler shows up as a large local_s, an idle rank as a large gather_s."""
    import time
    import torch.distributed as dist
    sizes = [np.asarray(s[0]).shape[0] for s in structures]
    t0 = time.perf_counter()
    if not (dist.is_available() and dist.is_initialized()):
        local = forward_local(forward_fn, structures, list(range(len(structures))), max_atoms)
        if timings is not None:
            timings.update(local_s=time.perf_counter() - t0, gather_s=0.0, structures=len(structures), atoms=int(sum(sizes)))
        return [local[i] for i in range(len(structures))]
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    mine = partition(sizes, world)[rank]
    # a rank whose structures ALL fail must still enter the collective (the others are already waiting in it): the verdict is taken
    # from the gathered descriptors, identically on every rank
    local = forward_local(forward_fn, structures, mine, max_atoms, raise_if_all_failed=False)
    t1 = time.perf_counter()
    failed = []
    out = gather_results(local, len(structures), n_out, group=group, device=device, failed_ranks=failed)
    if timings is not None:
        timings.update(local_s=t1 - t0, gather_s=time.perf_counter() - t1, structures=len(mine), atoms=int(sum(sizes[i] for i in mine)))
    if failed:
        raise AllStructuresFailed(f"every structure of rank(s) {failed} failed (see that rank's log): a systemic problem, not a bad input")
    return out


def broadcast_weights(model, state_dict=None, src=0, device=None):
    """SURVEY 8e "init: ncclBroadcast(weights)": rank ``src`` flattens its state_dict into the weight blob (pesto_amd.weights order), the
    blob travels through ONE torch.distributed broadcast - a device tensor under nccl (RCCL over xGMI), a host tensor under gloo - and every
    rank builds its handle from what it received (Model.load_blob); the other ranks need no checkpoint file. All ranks then compare a
    sha256 of their blob (all_gather_object): a rank that differs raises everywhere. Without a process group: plain load_state_dict.
    Returns {"bytes", "sha256_16", "ranks_equal", "backend"}."""
    import hashlib
    import torch
    import torch.distributed as dist
    from .weights import blob_size, flatten_state_dict
    n = blob_size(model.config)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        if state_dict is None:
            raise ValueError("broadcast_weights: no process group and no state_dict")
        model.load_state_dict(state_dict)
        h = hashlib.sha256(model.blob().tobytes()).hexdigest()[:16]
        return {"bytes": int(n) * 4, "sha256_16": h, "ranks_equal": True, "backend": None}
    backend = dist.get_backend()
    rank = dist.get_rank()
    on_dev = backend == "nccl"
    dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device()) if on_dev else torch.device("cpu"))
    if rank == src:
        if state_dict is None:
            raise ValueError(f"broadcast_weights: rank {src} needs the state_dict")
        blob = torch.from_numpy(np.ascontiguousarray(flatten_state_dict(model.config, state_dict), dtype=np.float32))
        t = blob.to(dev) if on_dev else blob
    else:
        t = torch.empty(n, dtype=torch.float32, device=dev if on_dev else "cpu")
    dist.broadcast(t, src=src)
    model.load_blob(t)
    h = hashlib.sha256(model.blob().tobytes()).hexdigest()[:16]
    hs = [None] * dist.get_world_size()
    dist.all_gather_object(hs, h)
    if len(set(hs)) != 1:
        raise RuntimeError(f"broadcast_weights: the ranks hold different weights after the broadcast: {hs}")
    return {"bytes": int(n) * 4, "sha256_16": h, "ranks_equal": True, "backend": backend}


def _parse_cpulist(s):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the kernel's cpulist format)"""
    out = []
    for part in s.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return out


def numa_node_of_pci(pci_bus_id, sysfs="/sys"):
    """NUMA node of a PCI device ('0000:0a:00' or '0000:0a:00.0'), or None where sysfs does not say (-1: a single-node host, a container
    without /sys/bus/pci)."""
    import os
    bdf = pci_bus_id if "." in pci_bus_id else pci_bus_id + ".0"
    try:
        node = int(open(os.path.join(sysfs, "bus", "pci", "devices", bdf.lower(), "numa_node")).read().strip())
    except (OSError, ValueError):
        return None
    return node if node >= 0 else None


def bind_rank_to_numa(device=None, sysfs="/sys", set_omp=True, max_threads=None):
    """Pin this process - one rank per GPU - to the CPUs of its GPU's NUMA node (the reference feeds its loop from
    DataLoader(num_workers=8), interfaceome/apply_model.py:49-50; here the rank's own threads pack and copy launches from host memory,
    0.85 ms of host CPU per structure: on a two-socket host a rank on the far socket pays the inter-socket link for every byte it packs
    and uploads). Reads /sys/bus/pci/devices/<bdf>/numa_node for the PCI bus id of the rank's device and /sys/devices/system/node/
    node<N>/cpulist, intersects with the CPUs the process may already use (a container's cpuset), calls os.sched_setaffinity and sets
    OMP_NUM_THREADS for libraries loaded afterwards. Returns {"numa_node", "cpus_allowed", "cpu_list_head", "bound", "why"}; binds nothing
    (bound False, with the reason) where the node is unknown - never an error: the placement is an optimisation."""
    import os
    info = {"numa_node": None, "cpus_allowed": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None,
            "cpu_list_head": None, "bound": False, "why": None}
    try:
        import torch
        if not torch.cuda.is_available():
            info["why"] = "no GPU"
            return info
        d = torch.cuda.current_device() if device is None else torch.device(device).index
        pr = torch.cuda.get_device_properties(d)
        bdf = f"{getattr(pr, 'pci_domain_id', 0):04x}:{getattr(pr, 'pci_bus_id', 0):02x}:{getattr(pr, 'pci_device_id', 0):02x}"
    except Exception as e:      # noqa: BLE001 - describing a device must not take the run down
        info["why"] = f"device properties unavailable: {e}"
        return info
    return _bind_to_node_of(bdf, info, sysfs, set_omp, max_threads)


def _bind_to_node_of(bdf, info, sysfs="/sys", set_omp=True, max_threads=None):
    import os
    node = numa_node_of_pci(bdf, sysfs)
    info["numa_node"] = node
    if node is None:
        info["why"] = f"{sysfs}/bus/pci/devices/{bdf}.0/numa_node gives no node (single-node host or no sysfs)"
        return info
    try:
        cpus = set(_parse_cpulist(open(os.path.join(sysfs, "devices", "system", "node", f"node{node}", "cpulist")).read()))
    except (OSError, ValueError) as e:
        info["why"] = f"no cpulist of node {node}: {e}"
        return info
    if hasattr(os, "sched_getaffinity"):
        cpus &= set(os.sched_getaffinity(0))      # (a container's cpuset: never ask for a CPU the process may not use)
    if not cpus:
        info["why"] = f"node {node} has no CPU this process may use"
        return info
    try:
        os.sched_setaffinity(0, cpus)
    except (OSError, AttributeError) as e:
        info["why"] = f"sched_setaffinity failed: {e}"
        return info
    n_thr = len(cpus) if max_threads is None else min(len(cpus), int(max_threads))
    if set_omp:
        os.environ["OMP_NUM_THREADS"] = str(n_thr)
    info.update({"cpus_allowed": len(cpus), "cpu_list_head": sorted(cpus)[:4], "bound": True, "omp_num_threads": n_thr})
    return info


def describe_ranks(device=None, numa=None):
    """What the collective library actually sees - for a bench line that must explain itself on hardware nobody could test on:
    ranks_seen = an all_reduce(SUM) of ones (device tensors under nccl: RCCL carried it), and per rank the device it computes on
    (name, PCI bus id, device ordinal, host; with ``numa`` = the dict bind_rank_to_numa returned: its NUMA node and the CPUs it is pinned
    to). {"world", "ranks_seen", "backend", "devices": [...]}; without a process group: world 1."""
    import socket
    import torch
    import torch.distributed as dist
    me = {"rank": 0, "host": socket.gethostname()}
    if torch.cuda.is_available():
        d = torch.cuda.current_device() if device is None else torch.device(device).index
        pr = torch.cuda.get_device_properties(d)
        me.update({"device": int(d), "device_name": pr.name, "pci_bus_id": f"{getattr(pr, 'pci_domain_id', 0):04x}:{getattr(pr, 'pci_bus_id', 0):02x}:{getattr(pr, 'pci_device_id', 0):02x}",
                   "gcn_arch": getattr(pr, "gcnArchName", None), "hbm_GiB": round(pr.total_memory / 2 ** 30, 1)})
    if numa is not None:
        me.update({"numa_node": numa.get("numa_node"), "cpus_allowed": numa.get("cpus_allowed"), "cpu_list_head": numa.get("cpu_list_head"),
                   "numa_bound": numa.get("bound"), "numa_why": numa.get("why")})
    if not (dist.is_available() and dist.is_initialized()):
        return {"world": 1, "ranks_seen": 1, "backend": None, "devices": [me]}
    backend = dist.get_backend()
    me["rank"] = dist.get_rank()
    one = torch.ones(1, dtype=torch.float32, device=(torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else "cpu"))
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    devs = [None] * dist.get_world_size()
    dist.all_gather_object(devs, me)
    return {"world": dist.get_world_size(), "ranks_seen": int(round(float(one.item()))), "backend": backend, "devices": devs,
            "distinct_devices": len({(d_.get("host"), d_.get("pci_bus_id"), d_.get("device")) for d_ in devs})}
