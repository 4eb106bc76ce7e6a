#!/bin/bash
# LDS bank conflicts of the layer kernels per library build: profiles/pmc_lds.sh <tag> [PESTO_LIB path]   (GPU box, repo root)
# one rocprofv3 --pmc pass (never combined with tracing); summary: python profiles/pmc_summary.py gpurun_out/pmc_<tag>
set -u
TAG=${1:-lds}
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; cd /tmp
[ $# -ge 2 ] && export PESTO_LIB=$2
mkdir -p $R/gpurun_out/pmc_$TAG
CMD="python $R/bench.py --steps 2 --warmup 1 --cpu-budget 0 --no-latency --no-extras --no-check --precision f16_split ${PESTO_BENCH_ARGS:-}"
timeout 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA \
  -d $R/gpurun_out/pmc_$TAG/p1 -o pmc --output-format csv -- $CMD > $R/gpurun_out/pmc_$TAG/p1.log 2>&1 || echo "pass failed"
