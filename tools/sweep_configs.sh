#!/bin/bash
# Measures the BASELINE.json parity-test configurations as bench lines (one JSON per line into gpurun_out/sweep_<tag>.jsonl).
#  config 3: block-size sweep at 131072 taps, 32768 samples per call, reference semantics (B >= 1024: the time-varying
#            reference behaviour in layered mode; also the exact linear convolution for comparison)
#  config 4: 64 streams, 524288-tap IR, uniform schedule and the native non-uniform one
#  config 5 per-GPU share: 1024 streams
TAG=${1:-r01}
OUT=gpurun_out/sweep_$TAG.jsonl
: > $OUT
for B in 128 256 512 1024 2048; do
  python bench.py --steps 8 --warmup 2 --no-cpu-baseline --block $B --blocks-per-call $((32768 / B)) >> $OUT 2>/dev/null
done
for B in 1024 2048; do
  python bench.py --steps 8 --warmup 2 --no-cpu-baseline --exact --block $B --blocks-per-call $((32768 / B)) >> $OUT 2>/dev/null
done
python bench.py --steps 8 --warmup 2 --no-cpu-baseline --streams 64 --ir-len 524288 >> $OUT 2>/dev/null
python bench.py --steps 8 --warmup 2 --no-cpu-baseline --streams 64 --ir-len 524288 --schedule nuc --blocks-per-call 512 >> $OUT 2>/dev/null
python bench.py --steps 8 --warmup 2 --no-cpu-baseline --streams 1024 >> $OUT 2>/dev/null
python - "$OUT" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    c = d["config"]
    print(c["streams_per_gpu"], c["ir_taps"], "B", c["block"], "T", c["blocks_per_call"], c["schedule"][:28], "->", d["value"], "M/s", d["ms_per_step"], "ms",
          {k: v for k, v in d["kernels_ms_per_step"].items() if v}, "mac GB/s", d["kernels"]["k_fdl_mac"]["achieved_gbs"])
PY
