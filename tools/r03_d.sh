#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
ROOT=$(pwd)
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/r03_d_trace -- python3 $ROOT/tools/train_latency.py 128 > $ROOT/gpurun_out/r03_d_trace.log 2>&1
cd $ROOT
tail -3 gpurun_out/r03_d_trace.log
T=$(ls gpurun_out/r03_d_trace/*/*kernel_trace.csv | head -1)
python tools/timeline.py $T gather_rows 700 40 > gpurun_out/r03_d_timeline_two_threads.txt; cat gpurun_out/r03_d_timeline_two_threads.txt
python tools/timeline.py $T gather_rows 100 24 > gpurun_out/r03_d_timeline_one_thread.txt; cat gpurun_out/r03_d_timeline_one_thread.txt
rm -rf gpurun_out/r03_d_trace
