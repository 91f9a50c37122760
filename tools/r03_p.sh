#!/bin/bash
mkdir -p gpurun_out
for cfg in "32 2 planes" "64 2 planes" "64 2 planes-device" "64 3 planes"; do
  set -- $cfg
  timeout -k 10 90 python tools/e2e_probe.py --agents $1 --predictors $2 --frames $3 --seconds 10 --warm 4 2>/dev/null | tail -1 > gpurun_out/r03_p_probe_$1_$2_$3.json
  python - gpurun_out/r03_p_probe_$1_$2_$3.json "$cfg" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[2], "| pps", d["predictions_per_sec"], "tps", d["train_steps_per_sec"], "batch", d["mean_predict_batch"], d["predictor_us_per_batch"], "| predict", d["engine"]["predict_us_per_call"], "| train", d["engine"]["train_us_per_call"], "| agent us/step", d["agent_cpu_us_per_step"], "server cores", d["server_cpu_cores"], d["cgroup"])
PY
done
