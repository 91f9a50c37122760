#!/bin/bash
# round 3, call t: can a small LDS-free kernel on a second stream run beside the fused conv kernels?
set -e
mkdir -p gpurun_out
{
for args in "128 64 64 2" "128 32 32 3" "64 32 32 3" "256 32 32 3" "32 64 64 3"; do
  timeout -k 10 120 ./tools/coresident $args
done
} > gpurun_out/r03_t_coresident.txt 2>&1
cat gpurun_out/r03_t_coresident.txt
