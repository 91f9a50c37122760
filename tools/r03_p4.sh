#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
{ GA3C_TIME_PREDICTIONS=1 timeout -k 10 120 python tools/predict_latency.py; timeout -k 10 120 python tools/predict_latency.py; timeout -k 10 100 python tools/ktime.py --batch 16 conv_stack_fwd dense1_fwd heads @predict; } > gpurun_out/r03_predict_latency.txt 2>&1
cat gpurun_out/r03_predict_latency.txt
