#!/bin/bash
# round 3, final build: the engine fed by native agent threads (tools/engine_ceiling.py) with rollouts that name their states
mkdir -p gpurun_out
for cfg in "256 2" "512 4" "64 2"; do
  set -- $cfg
  timeout -k 10 120 python tools/engine_ceiling.py --agents $1 --predictors $2 --seconds 12 --warm 4 2>/dev/null | tail -1 > gpurun_out/r03_ceil2_$1_$2.json
  python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r03_ceil2_%s_%s.json"%(sys.argv[1],sys.argv[2])).read())
print(sys.argv[1:], {k:d.get(k) for k in ('predictions_per_sec','train_steps_per_sec','mean_predict_batch','predictor_us_per_batch','threads_died')}, d['engine']['train_us_per_call'], d['cgroup'])
PY
done
