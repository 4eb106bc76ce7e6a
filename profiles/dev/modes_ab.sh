#!/bin/bash
# same box: the default build under pesto_debug_edge_mode 0 (chosen per launch), 1 (rendezvous), 2 (node waves); gpurun_out/modes/m<mode>_<rep>.json
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
rm -rf gpurun_out/modes; mkdir -p gpurun_out/modes
for rep in 1 2; do for m in 0 1 2; do
  timeout 300 python bench.py --batch ${B:-8} --steps 10 --warmup 3 --cpu-budget 0 --no-extras --no-latency --no-check --precision f16_split --edge-mode $m > gpurun_out/modes/m${m}_$rep.json 2>/dev/null
done; done
python profiles/ab_show.py gpurun_out/modes
