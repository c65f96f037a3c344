#!/bin/bash
# EQ alone at several stream counts: band-pipelined stages inside one launch (auto / forced count G) against one stage and
# against the event-ordered launches (G,N).  usage on the GPU box: bash tools/ab_stages.sh
cd ${GRAFT_REPO_ROOT:-$PWD}
for S in 1 4 16 64 128 256; do
  case $S in 1) L="auto 1 10 20 4,16";; 4) L="auto 1 5 10 20 4,16";; 16) L="auto 1 2 4 5 10 4,16";; 64) L="auto 1 2 4 5 4,16";; 128) L="auto 1 2 4,16";; *) L="auto 1";; esac
  for st in $L; do
    if [ "$st" = auto ]; then unset CPQ_SVF_STAGES; else export CPQ_SVF_STAGES=$st; fi
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity --eq-only --ir-len 4096 --streams $S --steps 8 --warmup 2 > /tmp/b.log 2>/dev/null || true
    python -c "
import json;d=json.loads(open('/tmp/b.log').read().strip().splitlines()[-1]);print('S=$S stages=$st', d['kernels_ms_per_step']['k_svf_cascade_tp'], 'ms')"
  done
done
