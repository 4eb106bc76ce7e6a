#!/bin/bash
# developer: registers / scratch of every kernel of a source file: profiles/dev/resources.sh [file.hip] [extra flags]
F=${1:-pesto_amd/csrc/pesto_edge.hip}; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -ffp-contract=on -x hip -c --cuda-device-only -Rpass-analysis=kernel-resource-usage "$@" $F -o /dev/null 2>&1 | python3 -c '
import re, sys
name = None
for l in sys.stdin:
    m = re.search(r"Function Name: (\S+)", l)
    if m: name = m.group(1); vg = ag = sc = occ = None
    m = re.search(r" VGPRs: (\d+)", l);  vg = m.group(1) if m else vg
    m = re.search(r"AGPRs: (\d+)", l);   ag = m.group(1) if m else ag
    m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", l); sc = m.group(1) if m else sc
    m = re.search(r"Occupancy \[waves/SIMD\]: (\d+)", l)
    if m:
        t = re.search(r"k_edgeILi(\d+)ELi(\d+)ELb(\d)ELi(\d+)ELi(\d+)E", name)
        short = ("k_edge<%s,%s,%s,%s,%s>" % t.groups()) if t else name[:40]
        print(f"{short:44s} VGPR {vg:>4s} AGPR {ag:>3s} scratch {sc:>4s} occ {m.group(1)}")
'
