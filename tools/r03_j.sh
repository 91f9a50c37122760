#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r03_j_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03_j_tests.log
( time timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r03_j_bench_k20.json 2> gpurun_out/r03_j_bench_k20.err ) 2>&1 | grep real
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03_j_bench_k20.json").read().strip().splitlines()[-1])
print("value %.3e ms/step %.5f wall %.5f"%(d["value"], d["ms_per_step"], d.get("ms_per_step_wall",0)))
print("lanes", {a:round(b/1e6,2) for a,b in d["predict_lanes"].items() if a in "1234"}, "8q", {a:round(b/1e6,2) for a,b in d.get("predict_lanes_8_hw_queues",{}).items() if a in "1234"})
print("train", d["train"]["ms_per_step"], d["train"].get("train_132",{}).get("ms_per_step"), "hogwild", d["train"].get("hogwild_lanes"))
print("roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_us"])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["train_steps_per_sec"])
e=d["e2e"]
print("e2e", round(e["predictions_per_sec"]), round(e["training_steps_per_sec"]), "x2", round(e["agents_x2"]["predictions_per_sec"]), round(e["agents_x2"]["training_steps_per_sec"]), "dev", round(e["agents_x2_frame_queue_on_device"]["predictions_per_sec"]), round(e["agents_x2_frame_queue_on_device"]["training_steps_per_sec"]))
for k,v in e["raw_frames"].items():
    if isinstance(v,dict): print("raw", k, {m: (round(r["predictions_per_sec"]), round(r["training_steps_per_sec"])) for m,r in v.items()})
print("config0", round(e["config0"]["predictions_per_sec"]), "cpu", round(d["cpu_baseline"]["e2e_config0"]["predictions_per_sec"]))
PY
