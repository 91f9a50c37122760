#!/bin/bash
mkdir -p gpurun_out
for cfg in "256 4 2" "256 4 4" "512 4 2" "256 2 2" "512 4 1"; do
  set -- $cfg
  GA3C_LANE_STREAMS=$3 timeout -k 10 120 python tools/engine_ceiling.py --agents $1 --predictors $2 --seconds 10 --warm 3 2>/dev/null | tail -1 > gpurun_out/r03_o_ceil_$1_$2_$3.json
  python - gpurun_out/r03_o_ceil_$1_$2_$3.json "$cfg" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print("agents pred lane_streams", sys.argv[2], "| pps", d["predictions_per_sec"], "tps", d["train_steps_per_sec"], "batch", d["mean_predict_batch"], d["predictor_us_per_batch"], "| predict", d["engine"]["predict_us_per_call"], "| train", d["engine"]["train_us_per_call"], "| cpu", d["cgroup"])
PY
done
