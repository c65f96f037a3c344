#!/bin/bash
# usage: tools/round_measure.sh <tag> [part]   (on the GPU box, from the repo root): everything a round's RESULTS / profiles quote.
#   part 1: the GPU test suite, the default bench line, the rocprofv3 passes of it (kernel stats, PMC traffic, SQ and LDS counters)
#   part 2: the configuration sweep, the streaming sweep, the clock probe
# (two gpurun calls: one call is limited to 20 minutes).  Outputs under gpurun_out/ (copy the summaries to profiles/).
TAG=$1
PART=${2:-all}
mkdir -p gpurun_out/$TAG
if [ "$PART" = 1 ] || [ "$PART" = all ]; then
  timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/$TAG/gputests.log 2>&1; tail -2 gpurun_out/$TAG/gputests.log
  timeout -k 10 300 python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err && echo "[round_measure] bench done"
  bash tools/profile_round.sh $TAG > gpurun_out/$TAG/profile_round.log 2>&1; tail -3 gpurun_out/$TAG/profile_round.log
fi
if [ "$PART" = 2 ] || [ "$PART" = all ]; then
  bash tools/sweep_configs.sh $TAG > gpurun_out/$TAG/sweep_configs.log 2>&1; tail -20 gpurun_out/$TAG/sweep_configs.log
  bash tools/sweep_streaming.sh $TAG > gpurun_out/$TAG/sweep_streaming.log 2>&1; tail -9 gpurun_out/$TAG/sweep_streaming.log
  bash tools/clock_probe.sh gpurun_out/$TAG/clock > gpurun_out/$TAG/clock_probe.txt 2>&1; cat gpurun_out/$TAG/clock_probe.txt
fi
