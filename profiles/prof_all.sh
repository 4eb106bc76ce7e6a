#!/bin/bash
# kernel-trace stats + PMC passes of the bench workload on the shipped build (GPU box, repo root)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/$TAG
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG/trace -o trace --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --cpu-budget 0 --no-latency --no-extras --precision f16_split > $R/gpurun_out/$TAG/trace.log 2>&1
cp $(find $R/gpurun_out/$TAG/trace -name "*kernel_stats.csv" | head -1) $R/gpurun_out/$TAG/kernel_stats.csv 2>/dev/null
cd $R
bash profiles/pmc_collect.sh $TAG > $R/gpurun_out/$TAG/pmc.log 2>&1
cp gpurun_out/pmc_$TAG/summary.txt gpurun_out/$TAG/pmc_summary.txt; cp gpurun_out/pmc_$TAG/traffic.json gpurun_out/$TAG/traffic.json
