#!/bin/bash
# hand-over between band-pipelined stages: agent-scope accesses against fences, for the loads (HL) and the stores (HS)
cd ${GRAFT_REPO_ROOT:-$PWD}
for v in "$@"; do
  rm -f convopeq_amd/csrc/build/svf_kernels.o
  make -C convopeq_amd/csrc EXTRA="$v" >/dev/null 2>&1
  for cfg in "1 10" "16 5" "64 4"; do
    set -- $cfg
    CPQ_SVF_STAGES=$2 timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity --eq-only --ir-len 4096 --streams $1 --steps 8 --warmup 2 > /tmp/b.log 2>/dev/null || true
    python -c "
import json;d=json.loads(open('/tmp/b.log').read().strip().splitlines()[-1]);print('[$v] S=$1 stages=$2', d['kernels_ms_per_step']['k_svf_cascade_tp'], 'ms')"
  done
done
rm -f convopeq_amd/csrc/build/svf_kernels.o; make -C convopeq_amd/csrc >/dev/null 2>&1
