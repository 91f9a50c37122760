#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r03_k_tests.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r03_k_tests.log
timeout -k 10 100 python tools/train_latency.py 128 132 2>&1 | grep "train_" | tee gpurun_out/r03_k_train_latency.txt
timeout -k 10 120 python tools/ktime.py --batch 128 @train @predict > gpurun_out/r03_k_ktime.txt 2>&1; cat gpurun_out/r03_k_ktime.txt
for cfg in "2 2 4" "4 4 4"; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$3 timeout -k 10 90 python tools/e2e_probe.py --agents 64 --predictors $1 --lanes $2 --seconds 10 --warm 4 2>/dev/null | tail -1 > gpurun_out/r03_k_probe_p$1_l$2_q$3.json
  python - gpurun_out/r03_k_probe_p$1_l$2_q$3.json "$cfg" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print("pred lanes queues", sys.argv[2], "| pps", d["predictions_per_sec"], "tps", d["train_steps_per_sec"], "batch", d["mean_predict_batch"], "| predict", d["engine"]["predict_us_per_call"], "| train", d["engine"]["train_us_per_call"], "reader waits", d["engine"]["train_reader_waits_per_call"], "| cpu", d["cgroup"])
PY
done
