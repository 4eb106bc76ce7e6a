#!/usr/bin/env python3
"""bench.py - structures/s of the PeSTo forward pass (i_v4_1 architecture) on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W   -> ONE JSON line on rank 0.
  * one "step" = one pass of Model.forward over one collated batch of --batch synthetic structures
    (N=3000 atoms, k=64, R=375 each; BASELINE.json configs[1]) with inputs already resident in HBM;
  * N>1: launched by torch.distributed.run, one rank per GPU; structures are independent, so each rank owns its
    own batch (weak scaling, no data-path collective); RCCL only carries the barrier and the max-over-ranks time;
  * value = structures processed by all ranks / max-over-ranks wall time of the K timed steps.
Extra objects: roofline (state-update kernels, HIP-event timed inside the library on the launch stream),
cpu_baseline (the C oracle = a port of the reference CPU path, timed on this host's cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: FP32 matrix = vector peak
PEAK_HBM_GBS = 8000.0     # HBM3E spec
PEAK_F16_TFLOPS = 2500.0  # dense f16 MFMA (the pipe the split GEMMs run on)


def layer_flops_per_atom(nn):
    """Reference-formulation FLOPs per atom per layer: 2*(13,696 + 36,376*n)."""
    return 2.0 * (13696.0 + 36376.0 * nn)


def executed_mfma_flops(config, n1):
    """FLOPs the MFMA pipes actually execute per forward on the shipped path (f16 hi/lo split: 3 products per GEMM).
    Edge kernel per 16-edge tile: 8 (16 when nn = 8) fp32 16x16x4 MFMAs (1,024 MAC) for the centre terms + 90 f16 16x16x32
    MFMAs (8,192 MAC: 24 for the per-edge p_j.r block of layer 1, 18 key networks, 48 value network); node kernel per 16
    atoms: 249 f16 MFMAs (the last launch only runs its 60-MFMA finish half)."""
    total = 0.0
    for l in config["sum"]:
        tiles = n1 * l["nn"] / 16.0
        total += tiles * ((16 if l["nn"] == 8 else 8) * 1024 + 90 * 8192) * 2.0
        total += n1 / 16.0 * 249 * 8192 * 2.0
    return total


def layer_gather_bytes_per_atom(nn):
    """Gather-counted bytes per atom per layer (definition A): 1,024 + 532*n."""
    return 1024.0 + 532.0 * nn


def make_batch(n_atoms, batch, seed0, n0, order="random"):
    from pesto_amd.topology import collate_batch_features, mask_to_segments, synthetic_structure
    items = [list(synthetic_structure(n_atoms, seed0 + b, n0=n0, order=order)) for b in range(batch)]
    X, ids, q, M = collate_batch_features(items)
    roa, R = mask_to_segments(M)
    return X, ids, q, roa, R


def load_weights(config):
    """i_v4_1 architecture with stacked real i_v4_0 weights (the trained i_v4_1 blob is absent upstream)."""
    from pesto_amd.weights import stack_layers, synthetic_state_dict
    path = os.path.join(ROOT, "tests", "golden", "weights_i_v4_0.npz")
    if len(config["sum"]) == 32 and os.path.exists(path):
        d = np.load(path)
        return stack_layers({k: d[k] for k in d.files}, config, 0.5), "stacked i_v4_0 (real) weights"
    return synthetic_state_dict(config, seed=0), "seeded random weights"


def cpu_baseline(config, sd, n_atoms, budget_s):
    """The C oracle (port of the reference CPU path, OpenMP over atoms) on this host's cores, bounded sample."""
    from oracle import oracle
    X, ids, q, roa, R = make_batch(n_atoms, 1, 1, config["em"]["N0"])
    m = oracle.OracleModel(config, sd)
    ids32 = ids.astype(np.int32)
    t0 = time.perf_counter()
    m.forward_segments(X, ids32, q, roa, R, stop_after=2)     # touch pages / spin up the OpenMP team
    t_warm = time.perf_counter() - t0
    times = []
    t_start = time.perf_counter()
    while not times or (time.perf_counter() - t_start + np.mean(times) < budget_s and len(times) < 5):
        t0 = time.perf_counter()
        m.forward_segments(X, ids32, q, roa, R)
        times.append(time.perf_counter() - t0)
    t = float(np.median(times))
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    omp = os.environ.get("OMP_NUM_THREADS")
    return {"value": 1.0 / t, "unit": "structures/s", "cores": int(omp) if omp else cores, "kind": "port",
            "sample": f"{len(times)} x one N={n_atoms} structure, all {config_name(config)} layers, C oracle (OpenMP); "
                      f"median {t:.2f} s/structure (2-layer warm-up {t_warm:.2f} s)"}


def config_name(config):
    return f"{len(config['sum'])}-layer"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="structures per step per GPU")
    ap.add_argument("--atoms", type=int, default=3000)
    ap.add_argument("--config", default="i_v4_1")
    ap.add_argument("--cpu-budget", type=float, default=20.0, help="seconds of CPU-baseline work (0 = skip)")
    ap.add_argument("--no-latency", action="store_true", help="skip the batch-1 latency side measurement")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend (nccl = RCCL)")
    ap.add_argument("--same-gpu", action="store_true",
                    help="testing only: every rank uses GPU 0 (lets the N>1 code path run on a 1-GPU box with --backend gloo)")
    ap.add_argument("--order", default="random", choices=["random", "morton"],
                    help="atom numbering of the synthetic clouds: generation order, or along a Z-order curve")
    args = ap.parse_args()

    import torch
    from pesto_amd import Model
    from pesto_amd.config import CONFIGS

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the forward pass)")
    gpu = 0 if args.same_gpu else local_rank
    torch.cuda.set_device(gpu)
    dev = torch.device("cuda", gpu)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    config = CONFIGS[args.config]
    n0 = config["em"]["N0"]
    sd, wdesc = load_weights(config)
    model = Model(config, validate=False).to(dev)
    model.load_state_dict(sd)

    # ---- inputs: one batch per rank, resident in HBM before the timed region
    X, ids, q, roa, R = make_batch(args.atoms, args.batch, 1000 * rank + 1, n0, args.order)
    Xd = torch.from_numpy(X).to(dev)
    idsd = torch.from_numpy(ids).to(dev)            # int64, as the reference passes it
    qd = torch.from_numpy(q).to(dev)
    road = torch.from_numpy(roa).to(dev)
    n_atoms_total = X.shape[0]

    def step():
        return model.forward_segments(Xd, idsd, qd, road, R)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        z = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        z = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(z).all()

    # ---- roofline leg: HIP events around the state-update launches, on the stream they run on
    model.set_timing(True)
    lay_ms = []
    for _ in range(max(5, min(args.steps, 20))):
        step()
        torch.cuda.synchronize()
        lay_ms.append(model.get_timing())
    # per-kernel pass (one event between consecutive layer launches): average launch duration of every kernel class
    model.set_timing(True, per_kernel=True)
    per_kernel = []
    for _ in range(5):
        step()
        torch.cuda.synchronize()
        per_kernel.append(model.get_kernel_timing())
    model.set_timing(False)
    kern = {}
    for name in per_kernel[0]:
        n_l = per_kernel[0][name][1]
        if n_l:
            kern[name] = {"launches_per_forward": n_l, "avg_launch_ms": float(np.median([pk[name][0] for pk in per_kernel])) / n_l}
    layers_ms = float(np.median([t["layers_ms"] for t in lay_ms]))
    fwd_all = np.array([t["total_ms"] for t in lay_ms])
    fwd_ms = float(np.median(fwd_all))
    n_launch = lay_ms[0]["n_layer_launches"]
    n1 = n_atoms_total + 1
    flops = sum(layer_flops_per_atom(l["nn"]) for l in config["sum"]) * n1
    gbytes = sum(layer_gather_bytes_per_atom(l["nn"]) for l in config["sum"]) * n1
    achieved_tf = flops / (layers_ms * 1e-3) / 1e12
    achieved_gbs = gbytes / (layers_ms * 1e-3) / 1e9
    # the dominant kernel on its own: k_edge<nn = max> (the edge part of a layer is 2 * 36,376 * nn FLOP and
    # 532 * nn gather-counted bytes per atom)
    nn_max = max(l["nn"] for l in config["sum"])
    dom = kern.get(f"edge_nn{nn_max}")
    dominant = None
    if dom:
        f_l = 2.0 * 36376.0 * nn_max * n1
        b_l = 532.0 * nn_max * n1
        t_all = sum(v["avg_launch_ms"] * v["launches_per_forward"] for v in kern.values())
        dominant = {"kernel": f"k_edge<{nn_max}>", "flops_per_launch": f_l, "avg_launch_ms": dom["avg_launch_ms"],
                    "achieved": f_l / (dom["avg_launch_ms"] * 1e-3) / 1e12, "unit": "TFLOP/s",
                    "frac": f_l / (dom["avg_launch_ms"] * 1e-3) / 1e12 / PEAK_F32_TFLOPS,
                    "gather_bytes_per_launch": b_l, "hbm_frac": b_l / (dom["avg_launch_ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS,
                    "share_of_layer_time": dom["avg_launch_ms"] * dom["launches_per_forward"] / t_all}
    # HBM-side bytes of the layer kernels from the committed PMC passes of this same command (profiles/pmc_collect.sh);
    # only quoted when the workload matches the one the counters were collected on
    traffic = None
    tpath = os.path.join(ROOT, "profiles", f"traffic_{args.config}_n{args.atoms}_b{args.batch}.json")
    if os.path.exists(tpath) and args.order == "random":
        traffic = json.load(open(tpath)).get("hbm_bytes_per_forward")

    # ---- side measurement: batch-1 latency (ms per structure when structures arrive one at a time)
    lat_ms = None
    if not args.no_latency and args.batch > 1:
        X1, ids1, q1, roa1, R1 = make_batch(args.atoms, 1, 1000 * rank + 1, n0, args.order)
        a = [torch.from_numpy(v).to(dev) for v in (X1, ids1, q1, roa1)]
        for _ in range(3):
            model.forward_segments(a[0], a[1], a[2], a[3], R1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(10):
            model.forward_segments(a[0], a[1], a[2], a[3], R1)
        torch.cuda.synchronize()
        lat_ms = (time.perf_counter() - t1) / 10 * 1e3

    if rank == 0:
        n_struct = args.steps * args.batch * world
        out = {
            "metric": "structures/sec (N=3000 atoms, k=64, 32 layers)" if args.config == "i_v4_1" and args.atoms == 3000
                      else f"structures/sec ({args.config}, N={args.atoms})",
            "value": n_struct / elapsed,
            "unit": "structures/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_structure": elapsed / n_struct * 1e3,
            "ms_per_structure_batch1": lat_ms,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.config} forward ({len(config['sum'])} state-update layers), synthetic cloud "
                                   f"N={args.atoms} atoms k=64 R={R // args.batch} per structure, {args.batch} structures "
                                   f"collated per step per GPU, atom order {args.order}, inputs resident in HBM, {wdesc}",
                       "atoms_per_step_per_gpu": int(n_atoms_total), "structures_per_step_per_gpu": args.batch,
                       "sharding": f"{world} rank(s), independent structures per rank, no data-path collective"},
            "roofline": {"bound": "mfma", "kernel": "state-update layer kernels (all launches of one forward)",
                         "achieved": achieved_tf, "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved_tf / PEAK_F32_TFLOPS, "traffic": traffic,
                         "flops_per_forward": flops, "launches": n_launch, "avg_launch_ms": layers_ms / n_launch,
                         "layers_ms": layers_ms, "forward_ms": fwd_ms,
                         "forward_ms_p10_p90": [float(np.percentile(fwd_all, 10)), float(np.percentile(fwd_all, 90))],
                         "executed_mfma_tflops": executed_mfma_flops(config, n1) / (layers_ms * 1e-3) / 1e12,
                         "executed_frac_of_f16_peak": executed_mfma_flops(config, n1) / (layers_ms * 1e-3) / 1e12 / PEAK_F16_TFLOPS,
                         "dominant_kernel": dominant, "kernels": kern,
                         "note": "achieved = reference-formulation FLOPs (SURVEY 8d) of all layer launches of one forward / their "
                                 "HIP-event time; peak = dense fp32 MFMA. The kernels execute ~2.5x fewer FLOPs (most of the first edge "
                                 "Linear folded per atom) and run the big GEMMs as f16 hi/lo split MFMA, so frac can exceed 1; "
                                 "traffic = HBM-side bytes per forward from rocprofv3 PMC (FETCH_SIZE x2 + WRITE_SIZE)"},
            "roofline_hbm": {"bound": "hbm", "achieved": achieved_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                             "frac": achieved_gbs / PEAK_HBM_GBS, "bytes_per_forward": gbytes,
                             "note": "gather-counted algorithmic bytes (SURVEY 8d definition A) / layer-kernel time"},
        }
        if args.cpu_budget > 0 and world == 1:      # the CPU baseline is a 1-GPU-run side measurement (rank 0, N = 1 only)
            out["cpu_baseline"] = cpu_baseline(config, sd, args.atoms, args.cpu_budget)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
