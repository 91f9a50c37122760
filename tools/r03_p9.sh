#!/bin/bash
# round 3: the answers of a batch given by a helper thread of the predictor loop -- engine A/B (64 agents), spin lengths
set -o pipefail
mkdir -p gpurun_out
{
for round in 1 2; do for cfg in "GA3C_RESPONDER=0" "GA3C_RESPONDER=1" "GA3C_RESPONDER=1 GA3C_RESPONDER_SPIN_US=0" "GA3C_RESPONDER=1 GA3C_RESPONDER_SPIN_US=60"; do
  echo "== engine, 64 agents, $cfg"
  env $cfg timeout -k 10 120 python tools/e2e_probe.py --agents 64 --seconds 8 --warm 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('predictions_per_sec','train_steps_per_sec','mean_predict_batch','predictor_us_per_batch','agent_wall_us_per_step','server_cpu_cores')}, d['cgroup'], d['engine']['predict_us_per_call'])"
done; done
echo "== 32 agents"
for cfg in "GA3C_RESPONDER=0" "GA3C_RESPONDER=1"; do
  env $cfg timeout -k 10 120 python tools/e2e_probe.py --agents 32 --seconds 8 --warm 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$cfg', {k:d[k] for k in ('predictions_per_sec','train_steps_per_sec','mean_predict_batch','predictor_us_per_batch','agent_wall_us_per_step','server_cpu_cores')}, d['cgroup'])"
done
} > gpurun_out/r03_p9.txt 2>&1
cat gpurun_out/r03_p9.txt
