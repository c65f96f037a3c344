#!/bin/bash
# as tools/ab_mac.sh, for one-block calls (the HBM-bound regime of the register-tile MAC variants)
for v in "$@"; do
  touch convopeq_amd/csrc/mac_kernels.hip
  make -C convopeq_amd/csrc EXTRA="$v" >/dev/null 2>&1 || { echo "build failed: $v"; continue; }
  python bench.py --blocks-per-call 1 --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms_per_step']
print('variant [$v]:', d['value'], 'M/s  mac', k['k_fdl_mac'], 'step', d['ms_per_step'])"
done
touch convopeq_amd/csrc/mac_kernels.hip
make -C convopeq_amd/csrc >/dev/null 2>&1
