#!/bin/bash
# round 3, call u: prediction lanes against the number of prediction streams at the runtime's default hardware queues
set -e
mkdir -p gpurun_out
out=gpurun_out/r03_u_lane_streams.txt
: > $out
for cfg in "GA3C_LANE_STREAMS=2" "GA3C_LANE_STREAMS=3" "GA3C_LANE_STREAMS=4" "GA3C_LANE_STREAMS=4 GPU_MAX_HW_QUEUES=8" "GA3C_LANE_STREAMS=4 GA3C_TRAIN_PRIORITY=0"; do
  echo "== $cfg" >> $out
  env $cfg timeout -k 10 120 python tools/lanes.py 128 6 >> $out 2>&1
done
cat $out
