#!/bin/bash
# A/B of compile-time variants of the vector-form EQ kernel: each argument is one set of -D flags (quote it)
# usage on the GPU box: bash tools/ab_tpv.sh "-DCPQ_TPV_PEAK=0" "-DCPQ_TPV_ORDER=3" ...
cd ${GRAFT_REPO_ROOT:-$PWD}
for v in "$@"; do
  rm -f convopeq_amd/csrc/build/svf_kernels.o
  make -C convopeq_amd/csrc EXTRA="$v" >/dev/null 2>&1
  for rep in 1 2; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity --eq-only --ir-len 4096 --steps 10 --warmup 3 > /tmp/b.log 2>/dev/null || true
  python -c "
import json;d=json.loads(open('/tmp/b.log').read().strip().splitlines()[-1]);print('variant [$v]', d['kernels_ms_per_step']['k_svf_cascade_tp'], 'ms')"
  done
done
rm -f convopeq_amd/csrc/build/svf_kernels.o; make -C convopeq_amd/csrc >/dev/null 2>&1
