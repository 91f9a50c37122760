#!/bin/bash
mkdir -p gpurun_out
for bw in 0 1; do
  echo "GA3C_TRAIN_BLOCKING_WAIT=$bw"
  GA3C_TRAIN_BLOCKING_WAIT=$bw timeout -k 10 100 python tools/train_latency.py 128 2>&1 | grep "train_offsets"
  for rep in 1 2; do
  GA3C_TRAIN_BLOCKING_WAIT=$bw timeout -k 10 90 python tools/e2e_probe.py --agents 64 --predictors 2 --seconds 10 --warm 4 2>/dev/null | tail -1 > gpurun_out/r03_l_probe_bw$bw_$rep.json
  python - gpurun_out/r03_l_probe_bw$bw_$rep.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print("  pps", d["predictions_per_sec"], "tps", d["train_steps_per_sec"], "| predict", d["engine"]["predict_us_per_call"], "| train", d["engine"]["train_us_per_call"], "| server cores", d["server_cpu_cores"], "agent cores", d["agent_cpu_cores"], d["cgroup"])
PY
  done
done
