#!/bin/bash
# Development aid: A/B of two builds of libga3c_hip.so on the same box, interleaved.
#   usage (through gpurun): bash tools/ab.sh "<ktime args>"   with tools/ab/lib_base.so and tools/ab/lib_new.so in place
set -e
for round in 1 2 3; do
  for v in base new; do
    cp tools/ab/lib_$v.so ga3c_amd/libga3c_hip.so
    echo "== $v (round $round)"
    timeout -k 10 120 python tools/ktime.py $1
  done
done
