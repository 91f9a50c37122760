#!/bin/bash
# round 3, call y: the whole -m gpu suite on the build with the dense1/w step inside conv_bwd, then the train figures
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r03_y_tests.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a gpurun_out/r03_y_tests.log; tail -4 gpurun_out/r03_y_tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 120 python tools/train_lanes.py 128 1 && timeout -k 10 120 python tools/train_lanes.py 121 1 && timeout -k 10 120 python tools/train_lanes.py 120 1 && timeout -k 10 120 python tools/train_lanes.py 132 1
