#!/bin/bash
# usage (GPU box, repo root): bash tools/clock_probe.sh <outdir> -- effective shader clock of every kernel of the default bench line
# and of the EQ alone: GRBM_GUI_ACTIVE (sum over the 8 XCDs) / 8 / kernel duration from the same pass's kernel trace.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/$1
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
run() {  # name, bench args
  n=$1; shift
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/$n -o c -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-parity "$@" > $O/$n.log 2>&1
  python3 - $O/$n <<'PY'
import csv, glob, sys, collections, re
d = sys.argv[1]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(cc)):
    m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
    if not m: continue
    agg[m.group(1)][r["Counter_Name"]].append((float(r["Counter_Value"]), dur[r["Dispatch_Id"]][0]))
for k, cs in sorted(agg.items()):
    g = cs.get("GRBM_GUI_ACTIVE")
    if not g: continue
    # skip warm-up launches: use the last 4
    g = g[-4:]
    clk = sum(v / 8.0 / t for v, t in g) / len(g)      # cycles per ns = GHz
    ms = sum(t for _, t in g) / len(g) / 1e6
    extra = {c: round(sum(v for v, _ in cs[c][-4:]) / len(cs[c][-4:]) / 1e6, 1) for c in cs if c != "GRBM_GUI_ACTIVE"}
    print(f"{sys.argv[1].split('/')[-1]:10s} {k:24s} {ms:8.3f} ms  clock {clk:5.3f} GHz  {extra}")
PY
}
run pipeline
run eqonly --eq-only --ir-len 4096
run eq1024 --eq-only --ir-len 4096 --streams 1024
run eq64 --eq-only --ir-len 4096 --streams 64
