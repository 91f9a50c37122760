#!/bin/bash
# round 3 experiment: the chip partitioned between predictions and training with CU masks (hipExtStreamCreateWithCUMask)
set -o pipefail
mkdir -p gpurun_out
{
for cfg in "X=0" "GA3C_PREDICT_CUS=64 GA3C_TRAIN_CUS_FROM=64" "GA3C_PREDICT_CUS=64" "GA3C_PREDICT_CUS=32 GA3C_TRAIN_CUS_FROM=32" "GA3C_TRAIN_CUS_FROM=64" "X=1"; do
  echo "== $cfg"
  env $cfg timeout -k 10 60 python tools/train_lanes.py 128 1
  env $cfg timeout -k 10 60 python tools/lanes.py 16 2
  env $cfg GA3C_TIME_PREDICTIONS=1 timeout -k 10 120 python tools/e2e_probe.py --agents 64 --seconds 8 --warm 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('predictions_per_sec','train_steps_per_sec','mean_predict_batch','predictor_us_per_batch','agent_wall_us_per_step')}, d['engine']['predict_us_per_call'], d['engine']['train_us_per_call'])"
done
} > gpurun_out/r03_cu_masks.txt 2>&1
cat gpurun_out/r03_cu_masks.txt
