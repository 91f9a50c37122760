#!/bin/bash
mkdir -p gpurun_out
for v in "0" "1"; do
  echo "TL_ROLLOUT_ROWS=1 GA3C_GATHER_DMA=$v"; TL_ROLLOUT_ROWS=1 GA3C_GATHER_DMA=$v timeout -k 10 100 python tools/train_latency.py 128 132 2>&1 | grep train_offsets | tee -a gpurun_out/r03_g_dma.txt
done
