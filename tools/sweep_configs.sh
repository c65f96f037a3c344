#!/bin/bash
# Measures the BASELINE.json configurations and the SURVEY 8(d) variants as bench lines (one JSON per line into
# gpurun_out/sweep_<tag>.jsonl; copy to profiles/).  524288 samples per call unless noted (the reference's largest block).
#  config 2 variants: saturation 0.0, one shared stereo IR, the AutoEq preset, a hot signal (the EQ's guarded output
#            stage), convolver only, and the round-1 schedule (P = 512, 64 blocks per call)
#  config 3: block-size sweep at 131072 taps, reference semantics (B >= 1024: the time-varying reference behaviour in
#            layered mode, at the engine's own P = 4096 and at P = B; also the exact linear convolution at P = 4096)
#  config 4: 64 streams, 524288-tap IR, uniform schedule and the reference's own non-uniform one
#  config 5 per-GPU share: 1024 streams
TAG=${1:-r02}
OUT=gpurun_out/sweep_$TAG.jsonl
: > $OUT
run() { python bench.py --steps 8 --warmup 2 --no-cpu-baseline "$@" >> $OUT 2>/dev/null || echo "FAILED: $*"; echo "done: $*"; }
run --saturation 0.0
run --shared-ir
run --eq-preset autoeq
run --pcm-scale 64
run --no-eq
run --partition 512 --blocks-per-call 64
for B in 128 256; do run --block $B --blocks-per-call $((524288 / B)); done     # the engine picks P = 4096 (CPQ_PARTITION_AUTO)
for B in 1024 2048; do run --block $B --blocks-per-call $((524288 / B)); done                  # layered mode, the engine picks P = 4096
for B in 1024 2048; do run --block $B --partition 0 --blocks-per-call $((524288 / B)); done    # the same at P = B
for B in 1024 2048; do run --exact --block $B --partition 4096 --blocks-per-call $((524288 / B)); done
run --streams 64 --ir-len 524288
run --streams 64 --ir-len 524288 --schedule nuc
run --streams 1024
python - "$OUT" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    c = d["config"]
    print(c["streams_per_gpu"], c["ir_taps"], "B", c["block"], "T", c["blocks_per_call"], "P", c["partition"], c["schedule"][:24], "|", c["workload"][60:120], "->", d["value"], "M/s", d["ms_per_step"], "ms",
          {k: v for k, v in d["kernels_ms_per_step"].items() if v}, "parity", (d.get("parity") or {}).get("rms_err"))
PY
python tools/summarize_profiles.py --check $OUT || echo "[sweep] a roofline fraction above 1: accounting error in bench.py"
