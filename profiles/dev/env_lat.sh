#!/bin/bash
# one structure per call against HIP runtime switches (kernel arguments in device memory, ...): bash profiles/dev/env_lat.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/envlat
run() { tag=$1; shift
  for rep in 1 2; do
    env "$@" timeout 300 python bench.py --steps 5 --warmup 2 --cpu-budget 0 --no-extras > gpurun_out/envlat/${tag}_$rep.json 2>/dev/null
    python -c "import json;d=json.load(open('gpurun_out/envlat/${tag}_$rep.json'));print('$tag', $rep, 'batch1 ms', round(d['ms_per_structure_batch1'],4), ' step ms', round(d['ms_per_step'],3))"
  done; }
run default X=1
run kernarg0 HIP_FORCE_DEV_KERNARG=0
run kernarg1 HIP_FORCE_DEV_KERNARG=1
run default_b X=1
