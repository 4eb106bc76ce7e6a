#!/bin/bash
# per-kernel averages of the small kernels of the forward (embed, unpack, pool, mask) under tagged builds: bash profiles/dev/aux_trace.sh tagA tagB ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; cd /tmp
for t in "$@"; do
  rm -rf /tmp/aux_$t
  PESTO_LIB=$R/pesto_amd/csrc/libpesto_hip_$t.so timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/aux_$t -o tr --output-format csv -- python $R/bench.py --steps 10 --warmup 3 --cpu-budget 0 --no-latency --no-extras --no-check --precision f16_split > /tmp/aux_$t.log 2>&1
  python - $t $(find /tmp/aux_$t -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, re, sys
rows = {re.sub(r"\(.*", "", r["Name"]).split("::")[-1][:28]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(sys.argv[2]))}
print(f"{sys.argv[1]:8s} " + "  ".join(f"{k} {v:.1f}" for k, v in rows.items() if k.startswith(("k_pool", "k_embed", "k_unpack", "k_mask", "k_node16"))))
PY
done
