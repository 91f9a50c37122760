#!/bin/bash
# round 3, call w: dense1/w stepped inside conv_bwd -- parity suite, then the train step and the kernels under both placements
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_train_parity.py -m gpu -x -q > gpurun_out/r03_w_tests.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a gpurun_out/r03_w_tests.log; tail -5 gpurun_out/r03_w_tests.log
[ $rc -eq 0 ] || exit 1
{
for round in 1 2; do
  for v in 1 0; do
    echo "== GA3C_WD_STEP_IN_CONV_BWD=$v"
    GA3C_WD_STEP_IN_CONV_BWD=$v timeout -k 10 120 python tools/train_lanes.py 128 1
    GA3C_WD_STEP_IN_CONV_BWD=$v timeout -k 10 120 python tools/train_lanes.py 64 1
  done
done
timeout -k 10 120 python tools/ktime.py --batch 128 conv_bwd conv_bwd_wdstep dense1_bwd_tile @train
} > gpurun_out/r03_w_ab.txt 2>&1
cat gpurun_out/r03_w_ab.txt
