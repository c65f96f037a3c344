#!/bin/bash
# A/B timing of mac_kernels.hip / fft_kernels.hip build variants on the GPU box: tools/ab_mac.sh "<EXTRA flags A>" "<EXTRA flags B>" ...
for v in "$@"; do
  touch convopeq_amd/csrc/mac_kernels.hip convopeq_amd/csrc/fft_kernels.hip
  make -C convopeq_amd/csrc EXTRA="$v" >/dev/null 2>&1 || { echo "build failed: $v"; continue; }
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels_ms_per_step']; print('variant [$v]:', d['value'], 'M/s  mac', k['k_fdl_mac'], 'fft', k['k_rfft_fwd_ols'], k['k_rfft_inv_ols'], 'svf', k['k_svf_cascade_tp'], 'step', d['ms_per_step'])"
done
touch convopeq_amd/csrc/mac_kernels.hip convopeq_amd/csrc/fft_kernels.hip
make -C convopeq_amd/csrc >/dev/null 2>&1
