#!/bin/bash
# usage (GPU box, repo root): bash tools/wg_trace.sh <outdir under gpurun_out>
# Per-workgroup trace of the EQ span kernel (tools/variants/make_wg_trace.py): where and when every workgroup of the last
# launch ran, in the default pipeline, for the EQ alone, at 1024 streams and at 64 streams (chained spans).
# The tracked source is put back (and the library rebuilt from it) however the script ends.
O=gpurun_out/$1
mkdir -p $O
TARGET=convopeq_amd/csrc/svf_kernels.hip
KEEP=$(mktemp); cp "$TARGET" "$KEEP"
trap 'cp "$KEEP" "$TARGET"; rm -f "$KEEP"; make -C convopeq_amd/csrc >/dev/null 2>&1' EXIT
python tools/variants/make_wg_trace.py "$KEEP" "$TARGET" || exit 1
make -C convopeq_amd/csrc > $O/build.log 2>&1 || { echo "variant build failed"; tail -5 $O/build.log; exit 1; }
run() {
  l=$1; shift
  timeout -k 10 300 python tools/wg_trace_probe.py $l --steps 8 --warmup 3 "$@" 2>$O/$l.err | grep -v '^{' | tee -a $O/summary.txt
}
run pipeline &&
run eqonly --eq-only --ir-len 4096 &&
run eq1024 --eq-only --ir-len 4096 --streams 1024 &&
run eq64 --eq-only --ir-len 4096 --streams 64
