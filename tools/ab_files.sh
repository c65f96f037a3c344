#!/bin/bash
# A/B timing of versions of one kernel source on the GPU box (same box, alternating):
#   tools/ab_files.sh <rounds> convopeq_amd/csrc/svf_kernels.hip <version A> <version B> ...
# Prints the kernel times of the default bench per build, and of the bench with each extra argument string of AB_EXTRA_ARGS
# (separated by ';', e.g. AB_EXTRA_ARGS="--eq-only --ir-len 4096;--streams 64").  The tracked file is put back (and the library rebuilt from it)
# however the script ends.
ROUNDS=$1; TARGET=$2; shift 2
IFS=';' read -r -a AB_EXTRA <<< "${AB_EXTRA_ARGS:-}"
KEEP=$(mktemp); cp "$TARGET" "$KEEP"
trap 'cp "$KEEP" "$TARGET"; rm -f "$KEEP"; make -C convopeq_amd/csrc >/dev/null 2>&1' EXIT
for r in $(seq $ROUNDS); do
  for v in "$@"; do
    cp "$v" "$TARGET"
    make -C convopeq_amd/csrc >/dev/null 2>&1 || { echo "build failed: $v"; continue; }
    for extra in "" "${AB_EXTRA[@]}"; do
    python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity $extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v [$extra]:', d['value'], 'M/s', {k: v for k, v in d['kernels_ms_per_step'].items() if v}, 'step', d['ms_per_step'])"
    done
  done
done
