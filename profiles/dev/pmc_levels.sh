#!/bin/bash
# average time a vector-memory / LDS instruction stays in flight per layer kernel (SQ_INST_LEVEL_* / SQ_INSTS_*), SALU share
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/pq_lv
timeout 240 rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d /tmp/pq_lv -o pmc --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --cpu-budget 0 --no-latency --no-extras --no-check --precision f16_split > /tmp/pq_lv.log 2>&1
tail -2 /tmp/pq_lv.log
python - /tmp/pq_lv <<'PY'
import csv, glob, re, sys, collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        m=re.search(r"k_edge<(\d+), (\d+), (?:true|false), \d+, (\d+)()>", row["Kernel_Name"])
        if not m: continue
        k=f"k_edge<{m.group(1)},NE={m.group(3)}>"
        agg[k][row["Counter_Name"]]+=float(row["Counter_Value"]); cnt[(k,row["Counter_Name"])]+=1
for k in sorted(agg, key=lambda s:int(re.search(r"<(\d+)",s).group(1))):
    d={c:v/cnt[(k,c)] for c,v in agg[k].items()}
    print(k, "  ".join(f"{c} {v/1e6:.3f}M" for c,v in sorted(d.items())))
    print("   VMEM level/insts = %.0f   LDS level/insts = %.0f   SALU cycles/wave-cycles = %.3f  SALU insts per VMEM inst %.1f" % (d["SQ_INST_LEVEL_VMEM"]/d["SQ_INSTS_VMEM"], d["SQ_INST_LEVEL_LDS"]/d["SQ_INSTS_LDS"], d["SQ_INST_CYCLES_SALU"]/d["SQ_WAVE_CYCLES"], d["SQ_INSTS_SALU"]/d["SQ_INSTS_VMEM"]))
PY
