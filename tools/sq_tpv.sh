#!/bin/bash
# SQ counters of the EQ kernel alone for compile-time variants: each argument is one set of -D flags (quote it)
# usage on the GPU box: bash tools/sq_tpv.sh <outdir> "-DCPQ_TPV_PEAK=0" ...
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/$1; shift
mkdir -p $O
export TMPDIR=/tmp
i=0
for v in "$@"; do
  i=$((i+1))
  rm -f $R/convopeq_amd/csrc/build/svf_kernels.o
  make -C $R/convopeq_amd/csrc EXTRA="$v" >/dev/null 2>&1
  (cd /tmp && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq$i -o q -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-parity --eq-only --ir-len 4096 > $O/sq$i.log 2>&1)
  python3 $R/tools/summarize_sq.py $(find $O/sq$i -name "*counter_collection.csv" | head -1) > $O/sq$i.json
  python3 -c "
import json;d=json.load(open('$O/sq$i.json'));
for k,v in d.items():
  if 'svf' in k: print('[$v]', k, {c:round(x['mean_per_launch']/1e6,1) for c,x in v.items()})"
done
rm -f $R/convopeq_amd/csrc/build/svf_kernels.o; make -C $R/convopeq_amd/csrc >/dev/null 2>&1
