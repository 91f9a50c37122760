#!/bin/bash
# round 3: the width of the training gather against the engine's prediction rate (PCIe bursts beside the predictions' own reads)
set -o pipefail
mkdir -p gpurun_out
{
for round in 1 2; do for g in 32 16 8 4; do
  echo "== engine, 64 agents, GA3C_GATHER_BLOCKS=$g"
  GA3C_GATHER_BLOCKS=$g GA3C_TIME_PREDICTIONS=1 timeout -k 10 120 python tools/e2e_probe.py --agents 64 --seconds 8 --warm 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('predictions_per_sec','train_steps_per_sec','mean_predict_batch','predictor_us_per_batch','agent_wall_us_per_step')}, d['engine']['predict_us_per_call'], d['engine']['train_us_per_call'])"
done; done
echo "== no training"
GA3C_TIME_PREDICTIONS=1 timeout -k 10 120 python tools/e2e_probe.py --agents 64 --seconds 8 --warm 3 --no-train 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('predictions_per_sec','train_steps_per_sec','mean_predict_batch','predictor_us_per_batch','agent_wall_us_per_step')}, d['engine']['predict_us_per_call'])"
} > gpurun_out/r03_gather_vs_predictions.txt 2>&1
cat gpurun_out/r03_gather_vs_predictions.txt
