"""Print the A/B lines profiles/ab.sh left in gpurun_out/ab/: structures/s, ms per step, per-layer-class launch times."""
import glob
import json
import sys

for f in sorted(glob.glob((sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/ab") + "/*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:      # a build that crashed leaves an empty file
        print(f, "unreadable:", e)
        continue
    r = d["roofline"]
    k = d.get("whole_forward", {}).get("kernels", {})
    cls = "  ".join(f"{n.replace('edge_', '')} {v['avg_launch_ms'] * 1e3:6.1f}" for n, v in sorted(k.items()))
    print(f"{f.split('/')[-1]:28s} {d['value']:8.1f} /s  {d['ms_per_step']:.3f} ms | us per launch: {cls}")
