#!/bin/bash
# everything the bench line's roofline block cites, from ONE gpurun call on the tree as it is:  bash profiles/prof_round.sh r05
#   gpurun_out/<tag>/kernel_stats.csv  rocprofv3 --kernel-trace --stats of the bench workload
#   gpurun_out/<tag>/pmc_summary.txt, pmc_readable.txt, traffic.json (hash-stamped, with the trace averages)  separate --pmc passes
#   gpurun_out/<tag>/pmc_per_nn.txt    dynamic instruction counts per 16-edge tile and issue-slot share of every layer kernel
#   gpurun_out/<tag>/pmc_classes.json, issue_floor_table.md   dynamic instruction classes + clock per layer kernel; predicted vs measured launch times
#   gpurun_out/<tag>/bench.json        the default bench line of the same tree on the same box (quotes the files above once copied)
# copy to profiles/<tag>_* and profiles/traffic_i_v4_1_n3000_b8.json afterwards (profiles/install_round.sh <tag>)
set -u
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash profiles/prof_all.sh $TAG
python profiles/pmc_summary.py gpurun_out/pmc_$TAG > gpurun_out/$TAG/pmc_readable.txt 2>&1
GRAFT_REPO_ROOT=$R bash profiles/dev/pmc_quick.sh default > $R/gpurun_out/$TAG/pmc_per_nn.txt 2>&1
cd $R
# dynamic instruction classes + the clock under every layer kernel -> the issue-floor table (profiles/issue_floor.py)
GRAFT_REPO_ROOT=$R bash profiles/pmc_classes.sh $TAG > $R/gpurun_out/$TAG/pmc_classes.log 2>&1
cd $R
cp gpurun_out/$TAG/traffic.json profiles/traffic_i_v4_1_n3000_b8.json
# the exact fp32 kernels (PESTO_PRECISION_FP32, what "auto" repeats a structure on): their kernel trace goes into the same stamped file
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG/trace_fp32 -o trace --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --cpu-budget 0 --no-latency --no-extras --precision fp32 > $R/gpurun_out/$TAG/trace_fp32.log 2>&1 )
cp $(find gpurun_out/$TAG/trace_fp32 -name "*kernel_stats.csv" | head -1) gpurun_out/$TAG/kernel_stats_fp32.csv 2>/dev/null
python - gpurun_out/$TAG/kernel_stats_fp32.csv gpurun_out/$TAG/traffic.json <<'PY'
import csv, json, re, sys
t = json.load(open(sys.argv[2])); tr = {}
try:
    for row in csv.DictReader(open(sys.argv[1])):
        m = re.search(r"(k_edge<[^>]*>|k_node\b)", row["Name"])
        if m and ("false" in m.group(1) or m.group(1) == "k_node"):
            tr[m.group(1).replace(" ", "")] = {"calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"])}
except OSError:
    pass
t["rocprof_kernel_trace_fp32"] = {"kernels": tr, "how": "rocprofv3 --kernel-trace --stats of python bench.py --steps 5 --warmup 2 --precision fp32 (same gpurun call)"}
json.dump(t, open(sys.argv[2], "w"), indent=1)
PY
# first bench line (un-stamped issue-floor keys) -> the floor table -> written into the stamped file -> the bench line the round commits
timeout 600 python bench.py --no-extras --cpu-budget 0 --no-latency > gpurun_out/$TAG/bench_pre.json 2> gpurun_out/$TAG/bench_pre.err
python profiles/issue_floor.py gpurun_out/$TAG/classes gpurun_out/$TAG/bench_pre.json gpurun_out/$TAG/traffic.json > gpurun_out/$TAG/issue_floor_table.md 2>&1
cp gpurun_out/$TAG/traffic.json profiles/traffic_i_v4_1_n3000_b8.json
timeout 900 python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
tail -c 600 gpurun_out/$TAG/pmc_per_nn.txt
head -c 1500 gpurun_out/$TAG/bench.json
