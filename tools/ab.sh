#!/bin/bash
# Development aid: A/B of builds of libga3c_hip.so on the same box, interleaved.
#   usage (through gpurun): bash tools/ab.sh "<ktime args>" [variants...]   with tools/ab/lib_<variant>.so in place (default: base new)
set -e
ARGS=$1; shift
VARS=${@:-base new}
for round in 1 2 3; do
  for v in $VARS; do
    cp tools/ab/lib_$v.so ga3c_amd/libga3c_hip.so
    echo "== $v (round $round)"
    timeout -k 10 120 python tools/ktime.py $ARGS
  done
done
