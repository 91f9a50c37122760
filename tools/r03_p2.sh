#!/bin/bash
# round 3: stream priorities in the running engine (64 agents, 2 predictors, 2 trainers): who should win a freed CU?
set -o pipefail
mkdir -p gpurun_out
out=gpurun_out/r03_priorities.txt
: > $out
for round in 1 2; do
for cfg in "X=0" "GA3C_PREDICT_PRIORITY=1 GA3C_TRAIN_PRIORITY=0" "GA3C_PREDICT_PRIORITY=1" "GA3C_TRAIN_PRIORITY=0"; do
  echo "== $cfg" >> $out
  env $cfg timeout -k 10 120 python tools/e2e_probe.py --agents 64 --seconds 8 --warm 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:(round(v,1) if isinstance(v,float) else v) for k,v in d.items() if k in ('predictions_per_sec','training_steps_per_sec','mean_predict_batch','predictor_cycle_us','cpu_cores_used')}, d.get('engine_stats',{}))" >> $out 2>&1
  echo "progress $cfg"
done
done
cat $out
