#!/bin/bash
# round 6: write-through (sc1) record / state stores A/B (wt0 plain, wt1 records, wt2 records + state) with a parity check of wt2,
# then the memory-side ablations of round 4 re-run on the round-6 kernels (results wrong by construction: --no-check)
mkdir -p gpurun_out/r06g
PESTO_LIB=$PWD/pesto_amd/csrc/libpesto_hip_wt2.so timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "golden or config5 or odd_launch or forced_edge or determinism or batch_independent" 2>&1 | tail -5 > gpurun_out/r06g/pytest_wt2.txt
rm -rf gpurun_out/ab; bash profiles/ab.sh wt0 wt1 wt2
python profiles/ab_show.py > gpurun_out/r06g/ab_wt.txt
rm -rf gpurun_out/ab; bash profiles/ab.sh wt0 nogather noprepst noprep nonode memall
python profiles/ab_show.py > gpurun_out/r06g/ab_mem.txt
cat gpurun_out/r06g/pytest_wt2.txt gpurun_out/r06g/ab_wt.txt gpurun_out/r06g/ab_mem.txt
