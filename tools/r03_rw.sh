#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_engine_e2e.py tests/test_gpu_train_parity.py -m gpu -x -q > gpurun_out/r03_rw_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03_rw_tests.log
timeout -k 10 200 python tools/train_latency.py 128 132 2>&1 | grep "NAMED\|one thread"
for i in 1 2 3; do
  timeout -k 10 120 python tools/e2e_probe.py --agents 64 --seconds 8 --warm 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d.get(k) for k in ('predictions_per_sec','train_steps_per_sec','mean_predict_batch','agent_wall_us_per_step')}, d['engine']['predict_us_per_call'], d['engine']['train_us_per_call'], d['engine']['train_reader_waits_per_call'])"
done
