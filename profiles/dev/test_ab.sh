#!/bin/bash
# one gpurun call: the whole GPU suite on the default build, then a same-box A/B of tagged builds: bash profiles/dev/test_ab.sh tagA tagB ...
# ("default" = pesto_amd/csrc/libpesto_hip.so); results: gpurun_out/test.log, gpurun_out/ab/*.json
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/test.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/test.log
rm -rf gpurun_out/ab; mkdir -p gpurun_out/ab
for rep in 1 2; do for t in "$@"; do
  if [ "$t" = default ]; then unset PESTO_LIB; else export PESTO_LIB=$R/pesto_amd/csrc/libpesto_hip_$t.so; fi
  timeout 300 python bench.py --batch 8 --steps 10 --warmup 3 --cpu-budget 0 --no-extras --no-latency --no-check --precision f16_split > gpurun_out/ab/${t}_$rep.json 2>/dev/null
done; done
unset PESTO_LIB
python profiles/ab_show.py
