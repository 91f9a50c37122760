#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_engine_e2e.py tests/test_gpu_baseline_configs.py -m gpu -x -q 2>&1 | tail -3
for pipe in 1 0; do
for cfg in "32 2 planes" "64 2 planes" "64 2 planes-device"; do
  set -- $cfg
  GA3C_PIPE=$pipe timeout -k 10 90 python - $1 $2 $3 $pipe <<'PY' 2>/dev/null | tail -1
import sys, subprocess, os, json
a, p, fr, pipe = sys.argv[1:5]
sys.path.insert(0, os.getcwd())
import ga3c_amd
from Config import Config
Config.PIPELINED_PREDICTOR = pipe == "1"
sys.argv = ["e2e_probe.py", "--agents", a, "--predictors", p, "--frames", fr, "--seconds", "10", "--warm", "4"]
import runpy
import io, contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    runpy.run_path("tools/e2e_probe.py", run_name="__main__")
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print("pipelined", pipe, a, p, fr, "| pps", d["predictions_per_sec"], "tps", d["train_steps_per_sec"], "batch", d["mean_predict_batch"], d["predictor_us_per_batch"], "| predict", d["engine"]["predict_us_per_call"], "| agent us/step", d["agent_cpu_us_per_step"], d["cgroup"])
PY
done
done
