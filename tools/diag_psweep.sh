#!/bin/bash
for cfg in "512 64" "2048 256" "4096 256" "4096 512" "4096 1024"; do
  set -- $cfg
  python bench.py --steps 6 --warmup 2 --no-cpu-baseline --partition $1 --blocks-per-call $2 > gpurun_out/pp_$1_$2.json 2>gpurun_out/pp_$1_$2.err
  python - "$1" "$2" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/pp_{sys.argv[1]}_{sys.argv[2]}.json"))
print("P", sys.argv[1], "T", sys.argv[2], "value", d["value"], "ms", d["ms_per_step"], d["kernels_ms_per_step"], d["roofline"]["kernel"])
PY
done
