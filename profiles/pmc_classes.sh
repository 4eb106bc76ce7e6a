#!/bin/bash
# dynamic instruction CLASSES of the layer kernels (the input of profiles/issue_floor.py): three rocprofv3 --pmc passes of the bench workload
# usage (GPU box, repo root): bash profiles/pmc_classes.sh <tag>   -> gpurun_out/<tag>/pmc_classes.json
set -u
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/$TAG/classes
cd /tmp
CMD="python $R/bench.py --steps 2 --warmup 1 --cpu-budget 0 --no-latency --no-extras --no-check --precision f16_split"
i=0
for SET in \
  "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64" \
  "SQ_INSTS_VALU_FMA_F16 SQ_INSTS_VALU_ADD_F16 SQ_INSTS_VALU_MUL_F16 SQ_INSTS_VALU_MFMA_F16 SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU" \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA" \
  "GRBM_GUI_ACTIVE SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"
do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $SET -d $R/gpurun_out/$TAG/classes/p$i -o pmc --output-format csv -- $CMD > $R/gpurun_out/$TAG/classes/p$i.log 2>&1 || echo "pass $i failed"
done
python - "$R/gpurun_out/$TAG/classes" "$R/gpurun_out/$TAG/pmc_classes.json" <<'PY'
import csv, glob, json, re, sys, collections
root, out = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(root + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.search(r"k_edge<(\d+), (\d+), (?:true|false), \d+, (\d+)()>", row["Kernel_Name"])
        if not m: continue
        k = f"k_edge<{m.group(1)},NE={m.group(3)}>"
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
res = {k: {c: v / cnt[(k, c)] for c, v in d.items()} for k, d in agg.items()}
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
for k in sorted(res, key=lambda s: int(re.search(r"<(\d+)", s).group(1))):
    d = res[k]; nn = int(re.search(r"<(\d+)", k).group(1)); tiles = 24001 * nn / 16
    print(k, " ".join(f"{c.replace('SQ_INSTS_VALU_', '').replace('SQ_INSTS_', '')}={v / tiles:.1f}" for c, v in sorted(d.items()) if c.startswith("SQ_INSTS")))
PY
