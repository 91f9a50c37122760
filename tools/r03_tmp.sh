#!/bin/bash
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tl -o t -- python3 $GRAFT_REPO_ROOT/tools/train_latency.py 132 > $GRAFT_REPO_ROOT/gpurun_out/r03_tl.txt 2>&1
grep "batch\|one thread" $GRAFT_REPO_ROOT/gpurun_out/r03_tl.txt
f=$(find /tmp/tl -name "*kernel_stats.csv" | head -1); cut -d, -f1-4 "$f" | head -14
