#!/bin/bash
# round 6: GPU suite, mask-scan A/B (m0 plain loads, m1 streaming loads) on the reference-signature headline, the full bench line, host packing by NUMA placement
mkdir -p gpurun_out/r06f gpurun_out/ab
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/r06f/pytest.txt
for rep in 1 2; do for t in m0 m1; do
  PESTO_LIB=$PWD/pesto_amd/csrc/libpesto_hip_$t.so python bench.py --steps 10 --warmup 3 --cpu-budget 0 --no-extras --no-latency --precision f16_split > gpurun_out/ab/${t}_$rep.json 2>/dev/null
done; done
python profiles/ab_show.py > gpurun_out/r06f/ab.txt
timeout 900 python bench.py > gpurun_out/r06f/bench.json 2> gpurun_out/r06f/bench.err
timeout 300 python profiles/host_packing.py numa 8 4 > gpurun_out/r06f/host_packing_numa.json 2> gpurun_out/r06f/host_packing_numa.err
cat gpurun_out/r06f/pytest.txt gpurun_out/r06f/ab.txt; head -c 600 gpurun_out/r06f/bench.json
