#!/bin/bash
# round 6, first A/B call: GPU suite on the tree's default library, microbenchmarks, then base / node1 / item1 interleaved (profiles/ab.sh)
mkdir -p gpurun_out/r06c
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r06c/pytest.txt
( cd profiles/microbench && timeout 120 ./valu_cost > ../../gpurun_out/r06c/valu_cost.txt 2>&1; timeout 180 ./grid_barrier > ../../gpurun_out/r06c/grid_barrier.txt 2>&1 )
rm -rf gpurun_out/ab; bash profiles/ab.sh base node1 item1
python profiles/ab_show.py > gpurun_out/r06c/ab.txt
# the reference-signature leg with the new mask scan
python bench.py --steps 10 --warmup 3 --cpu-budget 0 --no-latency --no-check > gpurun_out/r06c/bench_full.json 2> gpurun_out/r06c/bench_full.err
cat gpurun_out/r06c/pytest.txt gpurun_out/r06c/ab.txt
