#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
ROOT=$(pwd)
timeout -k 10 400 python -m pytest tests/test_gpu_train_parity.py tests/test_gpu_engine_e2e.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r03_f_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03_f_tests.log
timeout -k 10 100 python tools/train_latency.py 128 132 2>&1 | grep "train_" | tee gpurun_out/r03_f_train_latency.txt
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/r03_f_trace -- python3 $ROOT/tools/train_latency.py 128 > $ROOT/gpurun_out/r03_f_trace.log 2>&1
cd $ROOT
T=$(ls gpurun_out/r03_f_trace/*/*kernel_trace.csv | head -1)
python tools/timeline.py $T gather_rows 700 30 > gpurun_out/r03_f_timeline_two_threads.txt; cat gpurun_out/r03_f_timeline_two_threads.txt
rm -rf gpurun_out/r03_f_trace
