#!/bin/bash
# usage: ab.sh tagA tagB ...  -> runs each twice interleaved on the same box
mkdir -p gpurun_out/ab
for rep in 1 2; do for t in "$@"; do
  PESTO_LIB=$PWD/pesto_amd/csrc/libpesto_hip_$t.so python bench.py --steps 10 --warmup 3 --cpu-budget 0 --no-extras --no-latency --no-check --precision f16_split > gpurun_out/ab/${t}_$rep.json 2>/dev/null
done; done
