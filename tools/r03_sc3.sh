#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 200 python tools/train_latency.py 128 132 > gpurun_out/r03_sc3_train_latency.txt 2>&1; echo "train_latency rc=$?"; cat gpurun_out/r03_sc3_train_latency.txt
timeout -k 10 300 python -m pytest tests/test_gpu_engine_e2e.py -m gpu -x -q > gpurun_out/r03_sc3_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03_sc3_tests.log
for v in 1 0 1 0; do
  timeout -k 10 120 python tools/e2e_probe.py --agents 64 --seconds 8 --warm 3 --state-cache $v 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d.get(k) for k in ('state_cache','predictions_per_sec','train_steps_per_sec','mean_predict_batch','agent_wall_us_per_step','agent_cpu_us_per_step','threads_died')}, d['engine']['train_us_per_call'])"
done
