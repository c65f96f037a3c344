#!/bin/bash
# usage: tools/sweep_streaming.sh <tag>   (on the GPU box, from the repo root)
# The streaming regimes as bench lines (256 streams, 131072 taps, conv + EQ): one 512-sample block per call on the uniform
# and on the reference's own non-uniform schedule, and CPQ_CALLS_ANY at 480- / 441-sample quanta (one and eight callbacks
# per call) -> profiles/<tag>_sweep_streaming.jsonl
TAG=$1
OUT=gpurun_out/${TAG}_sweep_streaming.jsonl
: > $OUT
run() {
  echo "== $*" >&2
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-parity "$@" 2>/dev/null | tail -1 >> $OUT
}
run --blocks-per-call 1 --steps 200 --warmup 20
run --blocks-per-call 1 --steps 200 --warmup 20 --schedule nuc
run --blocks-per-call 8 --steps 100 --warmup 10 --schedule nuc
run --call-mode any --block 480 --blocks-per-call 1 --steps 200 --warmup 20
run --call-mode any --block 480 --blocks-per-call 8 --steps 100 --warmup 10
run --call-mode any --block 441 --blocks-per-call 1 --steps 200 --warmup 20
run --call-mode any --block 441 --blocks-per-call 8 --steps 100 --warmup 10
python - <<PY
import json
for l in open("$OUT"):
    d = json.loads(l)
    c = d["config"]
    print(c.get("schedule", "")[:60], "|", d["call_mode"], "B", c.get("block"), "T", c.get("blocks_per_call"), "|", d["value"], "M/s", d["ms_per_step"], "ms/step",
          "scopes", d["kernel_scopes_per_step"], "host us", d["host_enqueue_us_per_step"], {k: v for k, v in d["kernels_ms_per_step"].items() if v})
PY
python tools/summarize_profiles.py --check $OUT || echo "[sweep] a roofline fraction above 1: accounting error in bench.py"
