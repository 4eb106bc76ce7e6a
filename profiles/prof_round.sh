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
timeout 600 python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
python profiles/issue_floor.py gpurun_out/$TAG/classes gpurun_out/$TAG/bench.json > gpurun_out/$TAG/issue_floor_table.md 2>&1
tail -c 600 gpurun_out/$TAG/pmc_per_nn.txt
head -c 1500 gpurun_out/$TAG/bench.json
