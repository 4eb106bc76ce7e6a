#!/bin/bash
# instruction-fetch side of the layer kernels: which SQ / SQC instruction-cache counters the part offers, and their values per layer kernel
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --list-avail > /tmp/avail.txt 2>&1
grep -oE "\b(SQC?_[A-Z0-9_]*(ICACHE|IFETCH|INST_CACHE|IFETCH_LEVEL|WAIT_IFETCH)[A-Z0-9_]*)\b" /tmp/avail.txt | sort -u > /tmp/ic_names.txt
echo "available:"; cat /tmp/ic_names.txt | tr '\n' ' '; echo
grep -oE "\bSQ_(INSTS_[A-Z_0-9]+|INST_LEVEL[A-Z_0-9]*|WAIT_INST_ANY|INST_CYCLES[A-Z_0-9]*|THREAD_CYCLES_VALU|IFETCH[A-Z_0-9]*)\b" /tmp/avail.txt | sort -u | tr '\n' ' '; echo
NAMES=$(head -6 /tmp/ic_names.txt | tr '\n' ' ')
rm -rf /tmp/pq_if
timeout 240 rocprofv3 --pmc $NAMES SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d /tmp/pq_if -o pmc --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --cpu-budget 0 --no-latency --no-extras --no-check --precision f16_split > /tmp/pq_if.log 2>&1
tail -3 /tmp/pq_if.log
python - /tmp/pq_if <<'PY'
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
PY
