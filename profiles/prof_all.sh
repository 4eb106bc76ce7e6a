#!/bin/bash
# kernel-trace stats + PMC passes of the bench workload on the shipped build (GPU box, repo root)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/$TAG
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG/trace -o trace --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --cpu-budget 0 --no-latency --no-extras --precision f16_split > $R/gpurun_out/$TAG/trace.log 2>&1
cp $(find $R/gpurun_out/$TAG/trace -name "*kernel_stats.csv" | head -1) $R/gpurun_out/$TAG/kernel_stats.csv 2>/dev/null
cd $R
bash profiles/pmc_collect.sh $TAG > $R/gpurun_out/$TAG/pmc.log 2>&1
cp gpurun_out/pmc_$TAG/summary.txt gpurun_out/$TAG/pmc_summary.txt; cp gpurun_out/pmc_$TAG/traffic.json gpurun_out/$TAG/traffic.json
# the kernel trace's average durations go into the same hash-stamped file: bench.py quotes roofline.frac_rocprof from it
python - gpurun_out/$TAG/kernel_stats.csv gpurun_out/$TAG/traffic.json <<'PY'
import csv, json, re, sys
stats, tpath = sys.argv[1], sys.argv[2]
t = json.load(open(tpath))
tr = {}
for row in csv.DictReader(open(stats)):
    m = re.search(r"k_edge<([^>]*)>", row["Name"])
    if m:
        tr["k_edge<" + m.group(1).replace(" ", "") + ">"] = {"calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"])}
    elif "k_node16" in row["Name"]:
        tr["pesto::k_node16"] = {"calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"])}
t["rocprof_kernel_trace"] = {"kernels": tr, "how": "rocprofv3 --kernel-trace --stats of python bench.py --steps 20 --warmup 5 --precision f16_split "
                                                    "(the same gpurun call as the PMC passes; the profiler adds ~5 % to a launch)"}
json.dump(t, open(tpath, "w"), indent=1)
PY
