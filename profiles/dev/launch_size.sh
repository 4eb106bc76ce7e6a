#!/bin/bash
# config-4 leg (64 real chains from host memory, sharding.forward_sharded) against the atoms per collated launch.  bash profiles/dev/launch_size.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/launch_size
for rep in 1 2; do for ma in 24576 36864 49152 73728; do
  timeout 300 python bench.py --steps 5 --warmup 2 --cpu-budget 0 --no-latency --config4-max-atoms $ma > gpurun_out/launch_size/m${ma}_$rep.json 2>/dev/null
  python -c "
import json;d=json.load(open('gpurun_out/launch_size/m${ma}_$rep.json'));c=d['config4_sharded']
print('max_atoms', $ma, 'rep', $rep, ' compact', round(c['value'],1), ' dense', round(c['dense_forms']['value'],1), 'structures/s; headline', round(d['value'],1))"
done; done
