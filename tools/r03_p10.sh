#!/bin/bash
# round 3: the helper thread in the frames serve loop (frame queue on the device) -- tests, engine A/B
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_engine_e2e.py tests/test_gpu_frontend.py -m gpu -x -q > gpurun_out/r03_p10_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03_p10_tests.log
{
for round in 1 2; do for v in 1 0; do
  for fr in planes-device rgb-device; do
  GA3C_RESPONDER=$v timeout -k 10 120 python tools/e2e_probe.py --agents 64 --seconds 8 --warm 3 --frames $fr 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$fr helper=$v', {k:d.get(k) for k in ('predictions_per_sec','train_steps_per_sec','mean_predict_batch','predictor_us_per_batch','agent_wall_us_per_step','threads_died')})"
  done
done; done
} > gpurun_out/r03_p10.txt 2>&1
cat gpurun_out/r03_p10.txt
