#!/bin/bash
# round 3: conv2 of the fused conv stack with its K split over wave pairs -- parity, then A/B against the build before
set -o pipefail
mkdir -p gpurun_out
cp tools/ab/lib_new.so ga3c_amd/libga3c_hip.so
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_train_parity.py -m gpu -x -q > gpurun_out/r03_c2_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r03_c2_tests.log
[ $rc -eq 0 ] || exit 1
bash tools/ab.sh "--batch 128 conv_stack_fwd conv_stack_fwd_train conv_stack_fwd_u8 @predict @train" base new > gpurun_out/r03_c2_ab.txt 2>&1
cat gpurun_out/r03_c2_ab.txt
