#!/bin/bash
# Stall attribution for the edge kernel: separate rocprofv3 --pmc passes (never combined with tracing).
# The TA_* / TD_* counter sets are deliberately absent: collecting them aborted rocprofv3 and hung the run on this pool.
# usage (GPU box, repo root): bash profiles/pmc_stalls.sh <tag>
set -u
TAG=${1:-stalls}
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; cd /tmp
OUT=$R/gpurun_out/pmc_$TAG; mkdir -p $OUT
CMD="python $R/bench.py --steps 2 --warmup 1 --cpu-budget 0 --no-latency"
rocprofv3 -L > $OUT/avail.txt 2>&1 || true
i=0
for SET in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
  "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL" \
  "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
  "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
  "TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum"
do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $SET -d $OUT/p$i -o pmc --output-format csv -- $CMD > $OUT/p$i.log 2>&1 || echo "pass $i failed: $SET"
done
python - "$OUT" <<'PY'
import csv, glob, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(root + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][-36:]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
with open(root + "/summary.txt", "w") as out:
    for k in sorted(agg):
        if "k_edge" not in k and "k_node" not in k: continue
        print(k, file=out)
        for c in sorted(agg[k]): print(f"    {c:40s} {agg[k][c]/cnt[(k,c)]:18.1f}", file=out)
print(open(root + "/summary.txt").read())
PY
