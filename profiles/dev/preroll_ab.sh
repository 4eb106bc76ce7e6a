#!/bin/bash
# the driver's bench command with and without the clock pre-roll, alternating, same box.  bash profiles/dev/preroll_ab.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/preroll
for rep in 1 2 3; do for pm in 0 100; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-budget 0 --no-extras --preroll-ms $pm > gpurun_out/preroll/p${pm}_$rep.json 2>/dev/null
  python -c "
import json;d=json.load(open('gpurun_out/preroll/p${pm}_$rep.json'));b=d['batch1']
print('preroll_ms', $pm, 'rep', $rep, ' value', round(d['value'],1), 'structures/s  ms_per_step', round(d['ms_per_step'],4), ' 200-step median', round(d['long_sample']['ms_per_step_median'],4), ' batch1 steady', round(b['steady_state_ms'],4), 'cold burst', round(b['first_10_calls_after_100ms_idle_ms'],4), ' preroll steps', d['preroll']['steps'])"
done; done
