#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_train_parity.py tests/test_gpu_baseline_configs.py -m gpu -x -q > gpurun_out/r03_h_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03_h_tests.log
timeout -k 10 120 python tools/ktime.py --batch 128 @train conv_bwd dense1_bwd_tile conv_stack_fwd_train heads slab_reduce > gpurun_out/r03_h_ktime.txt 2>&1; cat gpurun_out/r03_h_ktime.txt
timeout -k 10 100 python tools/train_latency.py 128 132 2>&1 | grep "train_" | tee gpurun_out/r03_h_train_latency.txt
