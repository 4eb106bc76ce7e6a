#!/bin/bash
# round 6: timeline of developer builds (profiles/dev/timeline.py) + same-box A/B:  bash profiles/dev/r06_ab6.sh "tl tlp3" p0 p3
mkdir -p gpurun_out/r06i
for t in $1; do echo "== $t"; PESTO_LIB=$PWD/pesto_amd/csrc/libpesto_hip_$t.so timeout 300 python profiles/dev/timeline.py 2>&1 | grep "nn =" | grep -v "mean end\|workgroup index"; done > gpurun_out/r06i/timeline.txt
shift
rm -rf gpurun_out/ab; bash profiles/ab.sh "$@"
python profiles/ab_show.py > gpurun_out/r06i/ab.txt
cat gpurun_out/r06i/timeline.txt gpurun_out/r06i/ab.txt
