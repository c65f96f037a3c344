#!/bin/bash
# like ab_tpv.sh, plus the hot-signal case (--pcm-scale 64: the general output stage inside the band loop)
cd ${GRAFT_REPO_ROOT:-$PWD}
for v in "$@"; do
  rm -f convopeq_amd/csrc/build/svf_kernels.o
  make -C convopeq_amd/csrc EXTRA="$v" >/dev/null 2>&1
  for extra in "" "--pcm-scale 64"; do
  for rep in 1 2; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity --eq-only --ir-len 4096 --steps 10 --warmup 3 $extra > /tmp/b.log 2>/dev/null || true
  python -c "
import json;d=json.loads(open('/tmp/b.log').read().strip().splitlines()[-1]);print('variant [$v] [$extra]', d['kernels_ms_per_step']['k_svf_cascade_tp'], 'ms')"
  done
  done
done
rm -f convopeq_amd/csrc/build/svf_kernels.o; make -C convopeq_amd/csrc >/dev/null 2>&1
