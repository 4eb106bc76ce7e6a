#!/bin/bash
# per-launch time of every layer kernel against the launch size (structures of 3,000 atoms per launch): the intercept is what a launch
# costs before / after its steady state (launch, weight fill, first gathers, drain of the last centres).  bash profiles/dev/fixed_cost.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/fixed
for b in 1 2 3 4 6 8 12 16 24; do
  timeout 300 python bench.py --batch $b --steps 10 --warmup 3 --cpu-budget 0 --no-extras --no-latency > gpurun_out/fixed/b$b.json 2>/dev/null
done
python - <<'PY'
import json, numpy as np
bs = [1, 2, 3, 4, 6, 8, 12, 16, 24]
rows = {}
for b in bs:
    try: d = json.load(open(f"gpurun_out/fixed/b{b}.json"))
    except Exception as e: print(b, "failed", e); continue
    wf = d.get("whole_forward", {})
    per = {int(k): v["avg_launch_ms"] * 1e3 for k, v in d["roofline"]["per_nn"].items()}
    rows[b] = (d["ms_per_step"], per)
    print(b, "structures/launch: step", round(d["ms_per_step"], 4), "ms ", {k: round(v, 2) for k, v in per.items()})
for nn in (8, 16, 32, 64):
    x = np.array([b for b in rows if b >= 4 and nn in rows[b][1]], float); y = np.array([rows[int(b)][1][nn] for b in x])
    if len(x) > 2:
        A = np.vstack([x, np.ones_like(x)]).T; s, c = np.linalg.lstsq(A, y, rcond=None)[0]
        print(f"nn {nn}: t = {c:.1f} us + {s:.2f} us per structure (fit over launches of >= 4 structures)")
PY
