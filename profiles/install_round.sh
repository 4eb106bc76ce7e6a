#!/bin/bash
# copy what profiles/prof_round.sh <tag> left under gpurun_out/<tag>/ into the tracked profiles/ directory
set -eu
TAG=${1:-r05}
S=gpurun_out/$TAG
cp $S/kernel_stats.csv profiles/${TAG}_kernel_stats.csv
cp $S/pmc_summary.txt profiles/${TAG}_pmc_summary.txt
cp $S/pmc_readable.txt profiles/${TAG}_pmc_readable.txt
cp $S/pmc_per_nn.txt profiles/${TAG}_pmc_per_nn.txt
cp $S/traffic.json profiles/${TAG}_traffic.json
cp $S/traffic.json profiles/traffic_i_v4_1_n3000_b8.json
[ -s $S/kernel_stats_fp32.csv ] && cp $S/kernel_stats_fp32.csv profiles/${TAG}_kernel_stats_fp32.csv
[ -s $S/bench.json ] && cp $S/bench.json profiles/${TAG}_bench.json
[ -s $S/pmc_classes.json ] && cp $S/pmc_classes.json profiles/${TAG}_pmc_classes.json
[ -s $S/issue_floor_table.md ] && cp $S/issue_floor_table.md profiles/${TAG}_issue_floor_table.md
python - <<PY
import json, sys
sys.path.insert(0, ".")
import bench
t = json.load(open("profiles/traffic_i_v4_1_n3000_b8.json"))
print("stamp", t["source_hash"][:12], "tree", bench.source_hash()[:12], "match", t["source_hash"] == bench.source_hash())
PY
