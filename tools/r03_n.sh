#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
ROOT=$(pwd)
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_train_parity.py -m gpu -x -q > gpurun_out/r03_n_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03_n_tests.log
timeout -k 10 120 python tools/ktime.py --batch 128 @train @predict conv_stack_fwd conv_stack_fwd_train conv_bwd > gpurun_out/r03_n_ktime.txt 2>&1; cat gpurun_out/r03_n_ktime.txt
export TMPDIR=/tmp; cd /tmp
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $ROOT/gpurun_out/r03_n_fetch -- python3 $ROOT/bench.py --steps 20 --warmup 2 --predictors 1 --no-lane-sweep --cpu-seconds 0 --e2e-seconds 0 > /dev/null 2> $ROOT/gpurun_out/r03_n_fetch.err
cd $ROOT
python - <<'PY'
import csv,glob,collections
f=glob.glob("gpurun_out/r03_n_fetch/*/*counter_collection.csv")[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"]=="FETCH_SIZE": d[r["Kernel_Name"].split("(")[0].replace("void ","").replace("ga3c::","")].append(float(r["Counter_Value"]))
for k,v in sorted(d.items()):
    if "conv_stack" in k or "conv_bwd" in k: print("%-45s FETCH_SIZE %8.0f KB -> x2 = %.2f MB per launch (%d launches)"%(k, sum(v)/len(v), 2*sum(v)/len(v)*1024/1e6, len(v)))
PY
rm -rf gpurun_out/r03_n_fetch
