#!/bin/bash
# A/B of developer builds on ONE box (boxes of the pool differ by several %): profiles/ab.sh [-b BATCH] tagA tagB ...
# runs bench.py with each pesto_amd/csrc/libpesto_hip_<tag>.so twice, interleaved; results in gpurun_out/ab/<tag>_<rep>.json
B=8
M=0
if [ "$1" = "-b" ]; then B=$2; shift 2; fi
if [ "$1" = "-m" ]; then M=$2; shift 2; fi      # -m MODE: pesto_debug_edge_mode for every run
mkdir -p gpurun_out/ab
for rep in 1 2; do for t in "$@"; do
  PESTO_LIB=$PWD/pesto_amd/csrc/libpesto_hip_$t.so python bench.py --batch $B --steps 10 --warmup 3 --cpu-budget 0 --no-extras --no-latency --no-check --precision f16_split --edge-mode $M > gpurun_out/ab/${t}_$rep.json 2>/dev/null
done; done
