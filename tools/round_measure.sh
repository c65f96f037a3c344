#!/bin/bash
# usage: tools/round_measure.sh <tag>   (on the GPU box, from the repo root): everything a round's DESIGN / profiles quote,
# in one call -- the GPU test suite, the default bench line, the rocprofv3 passes of it, the configuration sweep and the
# streaming sweep.  Outputs under gpurun_out/ (copy the summaries to profiles/).
TAG=$1
mkdir -p gpurun_out/$TAG
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/$TAG/gputests.log 2>&1; tail -2 gpurun_out/$TAG/gputests.log
timeout -k 10 500 python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err && echo "[round_measure] bench done"
bash tools/profile_round.sh $TAG > gpurun_out/$TAG/profile_round.log 2>&1; tail -3 gpurun_out/$TAG/profile_round.log
bash tools/sweep_configs.sh $TAG > gpurun_out/$TAG/sweep_configs.log 2>&1; tail -20 gpurun_out/$TAG/sweep_configs.log
bash tools/sweep_streaming.sh $TAG > gpurun_out/$TAG/sweep_streaming.log 2>&1; tail -9 gpurun_out/$TAG/sweep_streaming.log
