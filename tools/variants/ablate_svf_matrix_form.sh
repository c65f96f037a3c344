#!/bin/bash
# (round 2; the CPQ_ABL hooks are in tools/variants/svf_matrix_form.hip, which is no longer built)
# timing ablation of the matrix-form EQ kernel (results are wrong by construction): builds variants with phases removed
# (CPQ_ABL bits: 1 reduction, 2 scan, 4 T.x product, 8 output stage) and times the EQ alone on the bench workload.
# usage on the GPU box: bash tools/ablate_svf.sh
set -e
cd ${GRAFT_REPO_ROOT:-$PWD}
for abl in 0 1 2 4 8 15; do
  rm -f convopeq_amd/csrc/build/svf_kernels.o
  make -C convopeq_amd/csrc EXTRA=-DCPQ_ABL=$abl >/dev/null 2>&1
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity --eq-only --ir-len 4096 --steps 10 --warmup 3 > /tmp/b.log 2>/dev/null || true
  python -c "
import json;d=json.loads(open('/tmp/b.log').read().strip().splitlines()[-1]);print('abl', $abl, d['kernels_ms_per_step']['k_svf_cascade_tp'], 'ms')"
done
rm -f convopeq_amd/csrc/build/svf_kernels.o; make -C convopeq_amd/csrc >/dev/null 2>&1
