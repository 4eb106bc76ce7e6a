#!/bin/bash
# same-box A/B of developer builds over several batch shapes: profiles/ab_sizes.sh tagA tagB ...  -> gpurun_out/abs/<tag>_b<B>_n<N>.json
mkdir -p gpurun_out/abs
TAGS="$@"
for t in $TAGS; do for cfg in "8 3000" "1 20000" "5 2500" "16 3000" "1 3000" "2 3000"; do set -- $cfg
  PESTO_LIB=$PWD/pesto_amd/csrc/libpesto_hip_$t.so python bench.py --batch $1 --atoms $2 --steps 15 --warmup 3 --cpu-budget 0 --no-extras --no-latency --precision f16_split > gpurun_out/abs/${t}_b$1_n$2.json 2>/dev/null
done; done
