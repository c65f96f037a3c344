#!/bin/bash
# A/B of whole-file variants of svf_kernels.hip: each argument is a source file that replaces it for one build
# usage on the GPU box: bash tools/ab_tpv_files.sh convopeq_amd/csrc/svf_kernels.hip tools/variants/<other>.hip ...
cd ${GRAFT_REPO_ROOT:-$PWD}
cp convopeq_amd/csrc/svf_kernels.hip /tmp/svf_kernels_current.hip
for round in 1 2; do
for v in "$@"; do
  cp "$v" /tmp/variant.hip; cp /tmp/variant.hip convopeq_amd/csrc/svf_kernels.hip
  [ "$v" = convopeq_amd/csrc/svf_kernels.hip ] && cp /tmp/svf_kernels_current.hip convopeq_amd/csrc/svf_kernels.hip
  rm -f convopeq_amd/csrc/build/svf_kernels.o
  make -C convopeq_amd/csrc >/dev/null 2>&1
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity --eq-only --ir-len 4096 --steps 10 --warmup 3 > /tmp/b.log 2>/dev/null || true
  python -c "
import json;d=json.loads(open('/tmp/b.log').read().strip().splitlines()[-1]);print('eq-only  [$v]', d['kernels_ms_per_step']['k_svf_cascade_tp'], 'ms')"
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity --steps 10 --warmup 3 > /tmp/b.log 2>/dev/null || true
  python -c "
import json;d=json.loads(open('/tmp/b.log').read().strip().splitlines()[-1]);print('pipeline [$v]', d['kernels_ms_per_step']['k_svf_cascade_tp'], 'ms', d['value'])"
done
done
cp /tmp/svf_kernels_current.hip convopeq_amd/csrc/svf_kernels.hip
rm -f convopeq_amd/csrc/build/svf_kernels.o; make -C convopeq_amd/csrc >/dev/null 2>&1
