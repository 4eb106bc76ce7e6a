set -u
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
mkdir -p $R/gpurun_out/pmc_e
for EB in 256 512; do
PESTO_EDGE_BLOCKS=$EB rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d $R/gpurun_out/pmc_e/eb$EB -o pmc --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --cpu-budget 0 --no-latency > $R/gpurun_out/pmc_e/eb$EB.log 2>&1
done
python - "$R/gpurun_out/pmc_e" <<'PY'
import csv, glob, sys, collections
root = sys.argv[1]
for eb in ("eb256", "eb512"):
    agg = collections.defaultdict(float); cnt = collections.Counter()
    for f in glob.glob(root + f"/{eb}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_edge<64" in row["Kernel_Name"]:
                agg[row["Counter_Name"]] += float(row["Counter_Value"]); cnt[row["Counter_Name"]] += 1
    wc = agg["SQ_WAVE_CYCLES"]
    print(eb, "k_edge<64>:", {k: round(v / wc, 3) for k, v in agg.items() if k != "SQ_WAVE_CYCLES"}, "wave_cycles/dispatch %.1fM" % (wc / cnt["SQ_WAVE_CYCLES"] / 1e6))
PY
