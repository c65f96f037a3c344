#!/bin/bash
# usage: tools/profile_round.sh <tag> [extra bench args]   (run on the GPU box from the repo root)
# rocprofv3 kernel stats + separate FETCH_SIZE / WRITE_SIZE PMC passes of the default bench, summarised into profiles/
set -e
TAG=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $O/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o f -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline "$@" > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o w -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline "$@" > $O/write.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/sq -o q -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline "$@" > $O/sq.log 2>&1 || true
find $O -name "*.csv" | head -20
