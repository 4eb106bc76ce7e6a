#!/bin/bash
# batch-1 latency (one N = 3000 structure per call) of tagged builds, same box: profiles/dev/lat_batch1.sh tagA tagB ...   ("default" = the shipped library)
for rep in 1 2; do for t in "$@"; do
  if [ "$t" = "default" ]; then unset PESTO_LIB; else export PESTO_LIB=$PWD/pesto_amd/csrc/libpesto_hip_$t.so; fi
  python bench.py --batch 1 --steps 50 --warmup 10 --cpu-budget 0 --no-extras --no-latency --precision ${PREC:-f16_split} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['whole_forward']['kernels']; print('$t', 'ms/step %.4f' % d['ms_per_step'], 'fwd_ms(events) %.4f' % d['whole_forward']['forward_ms'], 'layers %.4f' % d['whole_forward']['layers_ms'], ' '.join('%s %.1f' % (n.replace('edge_',''), v['avg_launch_ms']*1e3) for n,v in sorted(k.items())))"
done; done
