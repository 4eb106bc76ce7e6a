#!/bin/bash
# VGPR / scratch / LDS of every layer-kernel instantiation (compile-time resource report): profiles/regs.sh [extra hipcc flags]
cd "$(dirname "$0")/../pesto_amd/csrc" || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -ffp-contract=on "$@" -x hip -c ${PESTO_REGS_FILE:-pesto_edge.hip} -o /tmp/regs_probe.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|ScratchSize|LDS Size" | paste - - - - \
  | sed -E 's/remark: //g; s/\[-Rpass-analysis=kernel-resource-usage\]//g; s/pesto_[a-z_]+\.(hip|inc|h):[0-9]+:[0-9]+://g; s/Function Name: _ZN5pesto//; s/EvPKfNS_6LayerW[A-Za-z0-9_]*//'
