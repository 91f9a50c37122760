#!/bin/bash
# round 3, last build: the whole -m gpu suite, smoke, and the driver's bench command line
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r03_final_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r03_final_tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 60 python __graft_entry__.py smoke 2>&1 | tail -1
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r03_final_bench_B128_k20.json 2> gpurun_out/r03_final_bench_k20.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03_final_bench_B128_k20.json").read().strip().splitlines()[-1])
print("value %.3e ms/step %.5f"%(d["value"], d["ms_per_step"]), {a:round(b/1e6,2) for a,b in d["predict_lanes"].items() if a in "1234"}, "train", round(d["train"]["ms_per_step"],5), round(d["train"]["train_132"]["ms_per_step"],5), "roofline", round(d["roofline"]["frac"],4), d["roofline"]["traffic_commit"])
e=d["e2e"]; print("e2e", round(e["predictions_per_sec"]), round(e["training_steps_per_sec"]), "x2", round(e["agents_x2"]["predictions_per_sec"]), round(e["agents_x2"]["training_steps_per_sec"]), "dev", round(e["agents_x2_frame_queue_on_device"]["predictions_per_sec"]), round(e["agents_x2_frame_queue_on_device"]["training_steps_per_sec"]))
r=e["raw_frames"]; print({k:{kk:(round(vv["predictions_per_sec"]),round(vv["training_steps_per_sec"])) for kk,vv in v.items() if isinstance(vv,dict)} for k,v in r.items() if isinstance(v,dict)})
PY
