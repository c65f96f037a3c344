#!/bin/bash
# usage: tools/sweep_partition.sh <outdir>   (GPU box) -- throughput of configs[1] over FFT partition x blocks per call
O=${1:-gpurun_out/sweep_pt}; mkdir -p $O
for P in 512 1024 2048 4096; do
  for T in 64 128 256 512 1024; do
    python bench.py --partition $P --blocks-per-call $T --steps 8 --warmup 2 --no-cpu-baseline --no-parity > $O/P${P}_T$T.json 2> $O/P${P}_T$T.err || { echo "P $P T $T failed"; continue; }
    python - <<PY
import json
d=json.loads([l for l in open("$O/P${P}_T$T.json") if l.startswith("{")][-1])
k=d["kernels_ms_per_step"]; n=$T/64.0
print("P $P T $T: value", d["value"], "| per 64 blocks: fwd %.3f inv %.3f mac %.3f svf %.3f" % (k["k_rfft_fwd_ols"]/n, k["k_rfft_inv_ols"]/n, k["k_fdl_mac"]/n, k["k_svf_cascade_tp"]/n))
PY
  done
done
