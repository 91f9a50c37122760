#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_train_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_engine_e2e.py -m gpu -x -q > gpurun_out/r03_m_tests.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r03_m_tests.log
timeout -k 10 120 python tools/ktime.py --batch 128 @train @predict conv_stack_fwd conv_stack_fwd_train conv_stack_fwd_u8 > gpurun_out/r03_m_ktime.txt 2>&1; cat gpurun_out/r03_m_ktime.txt
timeout -k 10 60 python tools/lanes.py 128 4 2>&1 | tail -4
