#!/bin/bash
# same box: one-structure latency (bench.py's ms_per_structure_batch1, precision auto, device tensors) of tagged builds: bash profiles/dev/lat_ab.sh tagA tagB ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/lat
for rep in 1 2 3; do for t in "$@"; do
  if [ "$t" = default ]; then unset PESTO_LIB; else export PESTO_LIB=$R/pesto_amd/csrc/libpesto_hip_$t.so; fi
  timeout 300 python bench.py --steps 5 --warmup 2 --cpu-budget 0 --no-extras > gpurun_out/lat/${t}_$rep.json 2>/dev/null
  python -c "import json;d=json.load(open('gpurun_out/lat/${t}_$rep.json'));print('$t', $rep, 'batch1 ms', round(d['ms_per_structure_batch1'],4), ' step ms', round(d['ms_per_step'],3))"
done; done
