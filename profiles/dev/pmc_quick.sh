#!/bin/bash
# quick PMC: instruction counts per kernel for a tagged build
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
for t in "$@"; do
  LIBARG=""; [ "$t" != "default" ] && export PESTO_LIB=$R/pesto_amd/csrc/libpesto_hip_$t.so || unset PESTO_LIB
  rm -rf /tmp/pq_$t
  timeout 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES -d /tmp/pq_$t -o pmc --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --cpu-budget 0 --no-latency --no-extras --no-check --precision f16_split > /tmp/pq_$t.log 2>&1
  python - /tmp/pq_$t $t <<'PY'
import csv, glob, re, sys, collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        m=re.search(r"k_edge<(\d+), (\d+), (?:true|false), \d+, (\d+)()>", row["Kernel_Name"])
        if not m: continue
        k=f"k_edge<{m.group(1)},NE={m.group(3)}>"
        agg[k][row["Counter_Name"]]+=float(row["Counter_Value"]); cnt[(k,row["Counter_Name"])]+=1
for k in sorted(agg, key=lambda s:int(re.search(r"<(\d+)",s).group(1))):
    nn=int(re.search(r"<(\d+)",k).group(1)); tiles=24001*nn/16
    d={c:v/cnt[(k,c)] for c,v in agg[k].items()}
    print(f"{sys.argv[2]:8s} {k:18s} per tile: VALU {d['SQ_INSTS_VALU']/tiles:7.1f} MFMA {d['SQ_INSTS_MFMA']/tiles:6.1f} LDS {d['SQ_INSTS_LDS']/tiles:6.1f} VMEMrd {d['SQ_INSTS_VMEM_RD']/tiles:5.1f} SALU {d['SQ_INSTS_SALU']/tiles:6.1f} | active/wave-cycles {100*d['SQ_ACTIVE_INST_ANY']/d['SQ_WAVE_CYCLES']:.1f}% wavecycles {d['SQ_WAVE_CYCLES']/1e6:.1f}M")
PY
done
