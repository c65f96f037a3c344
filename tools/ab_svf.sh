#!/bin/bash
# A/B timing of svf_kernels.hip build variants on the GPU box: tools/ab_svf.sh "<EXTRA flags A>" "<EXTRA flags B>" ...
# (each argument is one variant; "" = the default build).  Prints the EQ kernel time of the default bench per variant.
for v in "$@"; do
  make -C convopeq_amd/csrc EXTRA="$v" >/dev/null 2>&1 || { echo "build failed: $v"; continue; }
  touch convopeq_amd/csrc/svf_kernels.hip
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('variant [$v]:', d['value'], 'M/s  svf', d['kernels_ms_per_step']['k_svf_cascade_tp'], 'ms  step', d['ms_per_step'])"
done
make -C convopeq_amd/csrc >/dev/null 2>&1
