#!/bin/bash
# EQ alone at S streams: event-ordered band-pipelined stages, G stages x N slices.  usage: bash tools/ab_stage_grid.sh S "G,N" ...
cd ${GRAFT_REPO_ROOT:-$PWD}
S=$1; shift
for st in "$@"; do
  for rep in 1 2; do
    CPQ_SVF_STAGES=$st timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity --eq-only --ir-len 4096 --streams $S --steps 8 --warmup 2 > /tmp/b.log 2>/dev/null || true
    python -c "
import json;d=json.loads(open('/tmp/b.log').read().strip().splitlines()[-1]);print('S=$S stages=$st', d['kernels_ms_per_step']['k_svf_cascade_tp'], 'ms')"
  done
done
