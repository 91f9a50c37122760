#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_engine_e2e.py tests/test_gpu_train_parity.py tests/test_gpu_baseline_configs.py -m gpu -x -q 2>&1 | tail -3
for cfg in "64 2 0" "64 2 4" "64 4 4"; do
  set -- $cfg
  L=""; if [ "$3" != "0" ]; then L="--lanes $3"; fi
  timeout -k 10 90 python tools/e2e_probe.py --agents $1 --predictors $2 $L --seconds 10 --warm 4 2>/dev/null | tail -1 > gpurun_out/r03_r_probe_$1_$2_$3.json
  python - gpurun_out/r03_r_probe_$1_$2_$3.json "$cfg" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print("agents pred lanes", sys.argv[2], "| pps", d["predictions_per_sec"], "tps", d["train_steps_per_sec"], "batch", d["mean_predict_batch"], "| predict", d["engine"]["predict_us_per_call"], "| train", d["engine"]["train_us_per_call"], "| cpu", d["cgroup"])
PY
done
timeout -k 10 60 python tools/lanes.py 128 4 2>&1 | tail -4
