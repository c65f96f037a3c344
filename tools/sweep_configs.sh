#!/bin/bash
# Measures the BASELINE.json parity-test configurations as bench lines (one JSON per line into gpurun_out/sweep_r01.jsonl).
#  config 3: block-size sweep at 131072 taps, 32768 samples per call (exact semantics for B >= 1024 where the
#            reference is time-varying, reference semantics otherwise)
#  config 4: 64 streams, 524288-tap IR
#  config 5 per-GPU share: 1024 streams
OUT=gpurun_out/sweep_r01.jsonl
: > $OUT
for B in 128 256 512; do
  python bench.py --steps 8 --warmup 2 --no-cpu-baseline --block $B --blocks-per-call $((32768 / B)) >> $OUT 2>/dev/null
done
for B in 1024 2048; do
  python bench.py --steps 8 --warmup 2 --no-cpu-baseline --exact --block $B --blocks-per-call $((32768 / B)) >> $OUT 2>/dev/null
done
python bench.py --steps 8 --warmup 2 --no-cpu-baseline --streams 64 --ir-len 524288 >> $OUT 2>/dev/null
python bench.py --steps 8 --warmup 2 --no-cpu-baseline --streams 1024 >> $OUT 2>/dev/null
python - <<'PY'
import json
for l in open("gpurun_out/sweep_r01.jsonl"):
    d = json.loads(l)
    c = d["config"]
    print(c["streams_per_gpu"], c["ir_taps"], "B", c["block"], "T", c["blocks_per_call"], "->", d["value"], "M/s", d["ms_per_step"], "ms",
          {k: v for k, v in d["kernels_ms_per_step"].items() if v}, "mac GB/s", d["kernels"]["k_fdl_mac"]["achieved_gbs"])
PY
