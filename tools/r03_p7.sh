#!/bin/bash
# round 3: the completion event of a prediction step as the stop event of its last launch -- tests, latency, engine A/B
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_engine_e2e.py tests/test_gpu_parity.py tests/test_gpu_train_parity.py tests/test_gpu_frontend.py -m gpu -x -q > gpurun_out/r03_p7_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r03_p7_tests.log
[ $rc -eq 0 ] || exit 1
{
for v in 1 0; do echo "== GA3C_STOP_EVENTS=$v"; GA3C_STOP_EVENTS=$v timeout -k 10 120 python tools/predict_latency.py; done
for round in 1 2; do for v in 1 0; do
  echo "== engine, 64 agents, GA3C_STOP_EVENTS=$v"
  GA3C_STOP_EVENTS=$v timeout -k 10 120 python tools/e2e_probe.py --agents 64 --seconds 8 --warm 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('predictions_per_sec','train_steps_per_sec','mean_predict_batch','predictor_us_per_batch','agent_wall_us_per_step')}, d['engine']['predict_us_per_call'])"
done; done
} > gpurun_out/r03_p7.txt 2>&1
cat gpurun_out/r03_p7.txt
