#!/bin/bash
# Per-kernel durations INSIDE a resident train step (rocprofv3 --kernel-trace --stats over tools/ktime.py @train), for a row
# count and two settings of the engine switches.   usage (through gpurun): bash tools/r04_step_trace.sh [rows]
set -o pipefail
B=${1:-132}
ROOT=$(pwd); export TMPDIR=/tmp; mkdir -p $ROOT/gpurun_out; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/trace_new -- python3 $ROOT/tools/ktime.py --batch $B --rounds 3 @train > $ROOT/gpurun_out/trace_new.txt 2>&1 || exit 1
GA3C_C2DW_OCC=2 GA3C_D1B_TAIL=0 GA3C_WD_STEP_IN_CONV2_DX=0 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/trace_old -- python3 $ROOT/tools/ktime.py --batch $B --rounds 3 @train > $ROOT/gpurun_out/trace_old.txt 2>&1 || exit 1
cd $ROOT
for v in new old; do
  echo "## $v (B = $B)"; tail -1 gpurun_out/trace_$v.txt
  python3 - $v <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/trace_%s/*/*kernel_stats.csv" % sys.argv[1])[0]
for r in csv.DictReader(open(f)):
    if int(r["Calls"]) >= 500: print("%-70s calls %5s avg %8.2f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
