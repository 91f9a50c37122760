#!/bin/bash
# A/B of two builds on prediction-lane throughput: bash tools/ab_lanes.sh [variants...]
set -e
VARS=${@:-base new}
for round in 1 2; do
  for v in $VARS; do
    cp tools/ab/lib_$v.so ga3c_amd/libga3c_hip.so
    echo "== $v (round $round)"
    GPU_MAX_HW_QUEUES=8 timeout -k 10 100 python tools/lanes.py 128 4
  done
done
