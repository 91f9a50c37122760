#!/bin/bash
mkdir -p gpurun_out
for g in 8 16 32 64 128 256; do
  echo "GA3C_GATHER_BLOCKS=$g"; GA3C_GATHER_BLOCKS=$g timeout -k 10 100 python tools/train_latency.py 128 132 2>&1 | grep train_offsets | tee -a gpurun_out/r03_e_gather_sweep.txt
done
