#!/bin/bash
# round 6: GPU suite on the tree's default library, then tagged builds interleaved (profiles/ab.sh): bash profiles/dev/r06_ab2.sh tagA tagB ...
mkdir -p gpurun_out/r06e
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/r06e/pytest.txt
rm -rf gpurun_out/ab; bash profiles/ab.sh "$@"
python profiles/ab_show.py > gpurun_out/r06e/ab.txt
cat gpurun_out/r06e/pytest.txt gpurun_out/r06e/ab.txt
