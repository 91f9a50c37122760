#!/bin/bash
# round 3: GPU span of the engine's prediction steps (timing events, GA3C_TIME_PREDICTIONS=1) with and without training beside them
set -o pipefail
mkdir -p gpurun_out
out=gpurun_out/r03_predict_span.txt
: > $out
for args in "--agents 64" "--agents 64 --no-train" "--agents 32" "--agents 64 --train-min-batch 122"; do
  echo "== $args" >> $out
  GA3C_TIME_PREDICTIONS=1 timeout -k 10 120 python tools/e2e_probe.py $args --seconds 8 --warm 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('predictions_per_sec','train_steps_per_sec','mean_predict_batch','predictor_us_per_batch','agent_wall_us_per_step','agent_cpu_us_per_step')}, d['engine'])" >> $out 2>&1
  echo "progress $args"
done
cat $out
