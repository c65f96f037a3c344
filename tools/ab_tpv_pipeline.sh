#!/bin/bash
# A/B of compile-time variants of the EQ kernel in the whole pipeline (default bench): step time, EQ time, and the EQ kernel's
# HBM traffic from two PMC passes.  usage on the GPU box: bash tools/ab_tpv_pipeline.sh "-DCPQ_TPV_PREFETCH=0" ...
cd ${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
for v in "$@"; do
  rm -f convopeq_amd/csrc/build/svf_kernels.o
  make -C convopeq_amd/csrc EXTRA="$v" >/dev/null 2>&1
  for rep in 1 2; do
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-parity --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('variant [$v]', d['value'], 'M/s step', d['ms_per_step'], 'EQ', d['kernels_ms_per_step']['k_svf_cascade_tp'])"
  done
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$c; (cd /tmp && rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- python3 ${GRAFT_REPO_ROOT:-/root/repo}/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity > /dev/null 2>&1)
    python3 - $c <<'PY'
import csv, glob, sys
c = sys.argv[1]
f = glob.glob(f"/tmp/pmc_{c}/**/*counter_collection.csv", recursive=True)
if f:
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if "k_svf_cascade_tpv" in r["Kernel_Name"] and r["Counter_Name"] == c]
    if v: print(f"   {c}: {sum(v)/len(v)*1024*(2 if c == 'FETCH_SIZE' else 1)/1e9:.3f} GB per launch ({len(v)} launches; FETCH doubled per the gfx950 rule)")
PY
  done
done
rm -f convopeq_amd/csrc/build/svf_kernels.o; make -C convopeq_amd/csrc >/dev/null 2>&1
