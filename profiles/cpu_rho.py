#!/usr/bin/env python3
"""SURVEY 8d step 1 (build container only): rho = t_reference / t_port per BASELINE config, both on the SAME cores.

  t_reference : the true reference Model.forward (PyTorch CPU, fp32, no_grad, eval, collated inputs), imported from
                /root/reference the way tests/golden/make_golden.py does; method of /root/reference/profiling.py:70-119
                (time around model(...)), but with 1 warm-up + >= 3 timed runs and the median reported;
  t_port      : the C oracle (oracle/pesto_oracle.c, OpenMP) on the same inputs and the same number of threads.

bench.py multiplies the port's time on the GPU box's host cores by rho to quote a reference-equivalent CPU time
(cpu_baseline.reference_equivalent); the table is copied into BASELINE.md.   Usage: python profiles/cpu_rho.py [--quick]
"""
import json
import os
import sys
import time

import numpy as np
import torch as pt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import make_golden as mg   # noqa: E402  (reference import helpers)


OUT_NAME = "r06_cpu_rho.json"      # (rounds 2 - 5: r02_cpu_rho.json, measured with the round-2 oracle)


def median_time(fn, runs):
    fn()
    ts = []
    for _ in range(runs):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), ts


def main():
    quick = "--quick" in sys.argv
    core = "--core" in sys.argv      # configs 1 - 3 only (the ones bench.py's reference_equivalent uses), three runs each, on a QUIET machine
    from conftest import weights
    from oracle import oracle
    from pesto_amd.config import CONFIGS
    from pesto_amd.topology import mask_to_segments
    threads = pt.get_num_threads()
    os.environ["OMP_NUM_THREADS"] = str(threads)
    import hashlib
    osrc = os.path.join(ROOT, "oracle", "pesto_oracle.c")
    out = {"threads": threads, "torch": pt.__version__, "oracle_source_sha16": hashlib.sha256(open(osrc, "rb").read()).hexdigest()[:16],
           "note": "rho is a property of (reference, port) on equal cores: bench.py refuses a file whose oracle_source_sha16 is not the hash of the "
                   "oracle source it times (VERDICT r5 item 8)", "configs": {}}

    cfg40, m40 = mg.load_run("i_v4_0_2021-09-07_11-20")
    cfg41, Model41, _ = mg.import_reference("i_v4_1_2021-09-07_11-21")
    m41 = Model41(cfg41).eval()
    m41.load_state_dict({k: pt.from_numpy(np.array(v)) for k, v in weights("i_v4_1").items()})
    cfg30, m30 = mg.load_run("i_v3_0_2021-05-27_14-27")
    mg.import_reference("i_v4_1_2021-09-07_11-21")

    def case(name, model, tag, inputs, runs):
        Xc, idsc, qc, Mc = mg.collate([list(inputs)])
        with pt.no_grad():
            t_ref, all_ref = median_time(lambda: model(Xc, idsc, qc, Mc.float()), runs)
        o = oracle.OracleModel(CONFIGS[tag], weights(tag))
        roa, R = mask_to_segments(Mc.numpy())
        X, ids, q = Xc.numpy(), idsc.numpy().astype(np.int32), qc.numpy()
        t_port, all_port = median_time(lambda: o.forward_segments(X, ids, q, roa, R), runs)
        out["configs"][name] = {"atoms": int(X.shape[0]), "t_reference_s": t_ref, "t_port_s": t_port, "rho": t_ref / t_port,
                                "runs_reference_s": all_ref, "runs_port_s": all_port}
        print(f"{name}: N={X.shape[0]} reference {t_ref:.2f} s, port {t_port:.2f} s, rho {t_ref / t_port:.1f}", flush=True)
        json.dump(out, open(os.path.join(ROOT, "profiles", OUT_NAME), "w"), indent=1)

    runs = 3
    st = mg.parse_pdb(os.path.join(mg.REF, "pdbs_test", "AY_2AYO_1_A:0.pdb"))
    case("1: i_v4_1 (stacked weights), pdbs_test 2AYO chain", m41, "i_v4_1", mg.encode(st, False), runs)
    case("2: i_v4_1, synthetic N=3000", m41, "i_v4_1", mg.synth_inputs(3000, 1), runs)
    case("3: i_v3_0, synthetic N=3000", m30, "i_v3_0", mg.synth_inputs(3000, 1, n0=123), runs)
    if core:
        return
    # config 4: the reference runs one chain per call; three chains spanning the size range stand for the 53
    for name in ("V9_2V9T_1_B:0", "WU_2WUS_1_A:0", "NV_3NVN_1_A:0"):
        st = mg.parse_pdb(os.path.join(mg.REF, "pdbs_test", name + ".pdb"))
        case(f"4: i_v4_1, pdbs_test {name}", m41, "i_v4_1", mg.encode(st, False), 3 if not quick else 1)
    if not quick:
        # topology from the package's k-d tree path (the reference's dense [N,N,3] extract_topology needs ~10 GB at N=20000;
        # the forward being timed is the reference's either way)
        from pesto_amd.topology import synthetic_structure
        X, ids0, q, M = synthetic_structure(20000, 5)
        case("5: i_v4_1, synthetic N=20000", m41, "i_v4_1",
             (pt.from_numpy(X), pt.from_numpy(ids0.astype(np.int64)), pt.from_numpy(q), pt.from_numpy(M)), 3)


if __name__ == "__main__":
    main()
