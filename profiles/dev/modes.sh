#!/bin/bash
# developer: bench the default library in the given pesto_debug_edge_mode values, one line per mode: profiles/dev/modes.sh 0 3 ...
mkdir -p gpurun_out/modes
for m in "$@"; do
  python bench.py --batch 8 --steps 10 --warmup 3 --cpu-budget 0 --no-extras --no-latency --precision f16_split --edge-mode $m > gpurun_out/modes/mode$m.json 2>/dev/null
done
python profiles/ab_show.py gpurun_out/modes
