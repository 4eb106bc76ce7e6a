#!/bin/bash
# same box, N interleaved repetitions of 20 steps, medians per tag:  bash profiles/dev/ab_n.sh N tagA tagB ...  (libpesto_hip_<tag>.so)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; rm -rf gpurun_out/ab; mkdir -p gpurun_out/ab
N=$1; shift
for rep in $(seq 1 $N); do for t in "$@"; do
  PESTO_LIB=$R/pesto_amd/csrc/libpesto_hip_$t.so timeout 300 python bench.py --batch 8 --steps 20 --warmup 5 --cpu-budget 0 --no-extras --no-latency --no-check --precision f16_split > gpurun_out/ab/${t}_$rep.json 2>/dev/null
done; done
python - "$@" <<'PY'
import glob, json, sys
import numpy as np
for t in sys.argv[1:]:
    rows = []
    for f in sorted(glob.glob(f"gpurun_out/ab/{t}_*.json")):
        try: d = json.load(open(f))
        except Exception: continue
        k = d["whole_forward"]["kernels"]
        rows.append([d["value"], d["ms_per_step"]] + [k[f"edge_nn{n}"]["avg_launch_ms"] * 1e3 for n in (8, 16, 32, 64)])
    if not rows: print(t, "no data"); continue
    a = np.array(rows); m = np.median(a, axis=0)
    print(f"{t:10s} n={len(rows)} median {m[0]:7.1f} /s {m[1]:.3f} ms | nn8 {m[2]:5.1f} nn16 {m[3]:5.1f} nn32 {m[4]:6.1f} nn64 {m[5]:6.1f} | all /s: " + " ".join(f"{v:.0f}" for v in a[:, 0]))
PY
