set -u
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
CMD="python $R/bench.py --steps 2 --warmup 1 --cpu-budget 0 --no-latency"
mkdir -p $R/gpurun_out/pmc_c
rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS_F32 SQ_IFETCH -d $R/gpurun_out/pmc_c/p1 -o pmc --output-format csv -- $CMD > $R/gpurun_out/pmc_c/p1.log 2>&1
rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 -d $R/gpurun_out/pmc_c/p2 -o pmc --output-format csv -- $CMD > $R/gpurun_out/pmc_c/p2.log 2>&1
python - "$R/gpurun_out/pmc_c" <<'PY'
import csv, glob, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(root + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][-30:]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
for k in sorted(agg):
    if "k_edge<64" not in k and "k_node" not in k: continue
    print(k)
    for c in sorted(agg[k]): print(f"    {c:30s} {agg[k][c]/cnt[(k,c)]:16.1f}")
PY
