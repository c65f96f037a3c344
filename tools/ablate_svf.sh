#!/bin/bash
# timing ablation of the MFMA span path (results are wrong by construction): builds variants with phases removed and
# times the EQ alone.  usage on the GPU box: bash tools/ablate_svf.sh
set -e
cd ${GRAFT_REPO_ROOT:-$PWD}
for abl in 0 1 2 4 8 15; do
  make -C convopeq_amd/csrc clean >/dev/null
  make -C convopeq_amd/csrc EXTRA=-DCPQ_ABL=$abl >/dev/null 2>&1
  for S in 128 256; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --eq-only --streams $S --ir-len 4096 > /tmp/b.log 2>&1 || true
    python -c "
import json;d=json.loads(open('/tmp/b.log').read().strip().splitlines()[-1]);print('abl', $abl, 'streams', $S, d['kernels_ms_per_step']['k_svf_cascade_tp'])"
  done
done
make -C convopeq_amd/csrc clean >/dev/null; make -C convopeq_amd/csrc >/dev/null 2>&1
