#!/bin/bash
# round 3, call x: per-kernel durations inside the train step under both placements of the dense1/w step (rocprofv3 --kernel-trace --stats)
set -o pipefail
mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  export GA3C_WD_STEP_IN_CONV_BWD=$v
  rm -rf /tmp/prof_$v
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o t -- python3 $R/tools/train_lanes.py 128 1 > $R/gpurun_out/r03_x_run_$v.log 2>&1 || exit 1
  f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1)
  cp "$f" $R/gpurun_out/r03_x_kernel_stats_wdstep_$v.csv
  echo "== GA3C_WD_STEP_IN_CONV_BWD=$v"; cut -d, -f1-4 "$f" | head -12
done
