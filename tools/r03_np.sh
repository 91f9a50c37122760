#!/bin/bash
# round 3, final build: predictor threads (NP) against the engine's rate, 64 Python agents
mkdir -p gpurun_out
for np_ in 2 3 4 2 3; do
  timeout -k 10 120 python tools/e2e_probe.py --agents 64 --predictors $np_ --seconds 8 --warm 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('NP=$np_', {k:d.get(k) for k in ('predictions_per_sec','train_steps_per_sec','mean_predict_batch','predictor_us_per_batch','agent_wall_us_per_step','server_cpu_cores')}, d['cgroup'], d['engine']['predict_us_per_call'])"
done
