#!/bin/bash
# usage: tools/profile_round.sh <tag> [extra bench args]   (run on the GPU box from the repo root)
# rocprofv3 kernel stats + separate FETCH_SIZE / WRITE_SIZE PMC passes of the default bench; the summaries (kernel stats
# csv, per-launch HBM bytes tagged with the workload and the kernel-source hash) land in gpurun_out/profiles_<tag>/ --
# copy them into profiles/ afterwards.
set -e
TAG=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-parity "$@" > $O/stats.log 2>&1
echo "[profile_round] stats pass done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o f -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-parity "$@" > $O/fetch.log 2>&1
echo "[profile_round] FETCH_SIZE pass done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o w -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-parity "$@" > $O/write.log 2>&1
echo "[profile_round] WRITE_SIZE pass done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O/sq -o q -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-parity "$@" > $O/sq.log 2>&1 || true
echo "[profile_round] SQ pass done"
# LDS pipe of every kernel (the P = 4096 FFT kernels are quoted as LDS-bandwidth bound): array cycles, conflict cycles, issue stalls
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/lds -o l -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-parity "$@" > $O/lds.log 2>&1 || true
echo "[profile_round] LDS pass done"
export CPQ_PROFILES_OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $CPQ_PROFILES_OUT
python3 tools/summarize_profiles.py $TAG "$(find $O/stats -name '*kernel_stats.csv' | head -1)" \
    "$(find $O/fetch -name '*counter_collection.csv' | head -1)" "$(find $O/write -name '*counter_collection.csv' | head -1)" \
    $O/stats.log > $O/summary.log 2>&1
SQ=$(find $O/sq -name '*counter_collection.csv' | head -1)
if [ -n "$SQ" ]; then python3 tools/summarize_sq.py "$SQ" $O/sq.log > $CPQ_PROFILES_OUT/${TAG}_sq_counters.json 2>> $O/summary.log || true; fi
LDS=$(find $O/lds -name '*counter_collection.csv' | head -1)
if [ -n "$LDS" ]; then python3 tools/summarize_sq.py "$LDS" $O/lds.log > $CPQ_PROFILES_OUT/${TAG}_lds_counters.json 2>> $O/summary.log || true; fi
grep -h '"metric"' $O/stats.log > $CPQ_PROFILES_OUT/${TAG}_bench_profiled.json || true
ls -la $CPQ_PROFILES_OUT
