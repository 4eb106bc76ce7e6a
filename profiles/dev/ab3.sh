#!/bin/bash
# same box, three interleaved repetitions of 20 steps: bash profiles/dev/ab3.sh tagA tagB ...  (libpesto_hip_<tag>.so)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; rm -rf gpurun_out/ab; mkdir -p gpurun_out/ab
for rep in 1 2 3; do for t in "$@"; do
  PESTO_LIB=$R/pesto_amd/csrc/libpesto_hip_$t.so timeout 300 python bench.py --batch 8 --steps 20 --warmup 5 --cpu-budget 0 --no-extras --no-latency --no-check --precision f16_split > gpurun_out/ab/${t}_$rep.json 2>/dev/null
done; done
python profiles/ab_show.py
