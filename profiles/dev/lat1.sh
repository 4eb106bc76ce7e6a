#!/bin/bash
# developer: kernel trace of one-structure forwards (N = 3000): kernel durations vs the gaps between them
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; cd /tmp
mkdir -p $R/gpurun_out/lat1
timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/lat1/trace -o trace --output-format csv -- python $R/bench.py --batch ${LAT_BATCH:-1} --steps 20 --warmup 5 --cpu-budget 0 --no-latency --no-extras --no-check --precision f16_split > $R/gpurun_out/lat1/log.txt 2>&1
python - "$R/gpurun_out/lat1" <<'PY'
import csv, glob, sys, re
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last complete forward: from a k_embed to the next k_pool_reduce
idx = [i for i, r in enumerate(rows) if "k_embed" in r["Kernel_Name"]]
a = idx[-2]; b = idx[-1]
seg = rows[a:b]
t0 = int(seg[0]["Start_Timestamp"])
tot_k = 0; tot_gap = 0; prev_end = None
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    m = re.search(r"k_edge<(\d+), (\d+)", r["Kernel_Name"])
    name = f"k_edge<{m.group(1)},{m.group(2)}>" if m else re.sub(r"\(.*", "", r["Kernel_Name"])[-28:]
    gap = (s - prev_end) if prev_end is not None else 0
    tot_k += e - s; tot_gap += gap
    print(f"{name:30s} start {(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:6.1f}  gap before {gap / 1e3:5.1f}  grid {r.get('Grid_Size', '?')} wg {r.get('Workgroup_Size', '?')}")
    prev_end = e
print(f"forward: {len(seg)} launches, kernels {tot_k / 1e3:.1f} us + gaps {tot_gap / 1e3:.1f} us = {(int(seg[-1]['End_Timestamp']) - t0) / 1e3:.1f} us")
PY
