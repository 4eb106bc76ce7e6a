#!/bin/bash
# same-box A/B under SUSTAINED load: the 200-step sample of bench.py (HIP events between consecutive steps), each library twice, interleaved
mkdir -p gpurun_out/abL
for rep in 1 2; do for t in "$@"; do
  PESTO_LIB=$PWD/pesto_amd/csrc/libpesto_hip_$t.so python bench.py --batch 8 --steps 20 --warmup 3 --cpu-budget 0 --no-latency --no-check --precision f16_split > gpurun_out/abL/${t}_$rep.json 2>/dev/null
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/abL/*.json")):
    d=json.load(open(f)); k=d["whole_forward"]["kernels"]; ls=d.get("long_sample") or {}
    print(f.split("/")[-1], "%.1f /s" % d["value"], "long-sample median %.3f ms" % ls.get("ms_per_step_median", float("nan")), " ".join("%s %.1f" % (n.replace("edge_",""), v["avg_launch_ms"]*1e3) for n,v in sorted(k.items())))
PY
