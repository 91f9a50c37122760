#!/bin/bash
# round 3, last call: the whole -m gpu suite and the driver's bench command line on the final commit
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r03_zz_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r03_zz_tests.log; tail -3 gpurun_out/r03_zz_tests.log
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r03_zz_bench_B128_k20.json 2> gpurun_out/r03_zz_bench_k20.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03_zz_bench_B128_k20.json").read().strip().splitlines()[-1])
print("value %.3e ms/step %.5f wall %.5f"%(d["value"], d["ms_per_step"], d.get("ms_per_step_wall",0)), {a:round(b/1e6,2) for a,b in d["predict_lanes"].items() if a in "1234"}, "train", round(d["train"]["ms_per_step"],5), round(d["train"]["train_132"]["ms_per_step"],5), "roofline", round(d["roofline"]["frac"],4), d["roofline"]["avg_launch_us"])
e=d["e2e"]; print("e2e", round(e["predictions_per_sec"]), round(e["training_steps_per_sec"]), "x2", round(e["agents_x2"]["predictions_per_sec"]), round(e["agents_x2"]["training_steps_per_sec"]), "dev", round(e["agents_x2_frame_queue_on_device"]["predictions_per_sec"]), round(e["agents_x2_frame_queue_on_device"]["training_steps_per_sec"]), "8q", {a:round(b/1e6,2) for a,b in d.get("predict_lanes_8_hw_queues",{}).items() if a in "1234"})
for k,v in e["raw_frames"].items():
    if isinstance(v,dict): print("raw", k, {m:(round(r["predictions_per_sec"]),round(r["training_steps_per_sec"])) for m,r in v.items()})
print("config0", round(e["config0"]["predictions_per_sec"]), round(d["cpu_baseline"]["e2e_config0"]["predictions_per_sec"]), "cpu", round(d["cpu_baseline"]["value"]), round(d["cpu_baseline"]["train_steps_per_sec"],1), "hog", {k:round(v) for k,v in d["train"]["hogwild_lanes"].items() if k in "124"}, "u8", round(d["uint8_resident"]["predictions_per_sec"]), round(d["uint8_resident"]["training_steps_per_sec"]))
PY
for cfg in "256 4" "512 4"; do
  set -- $cfg
  timeout -k 10 120 python tools/engine_ceiling.py --agents $1 --predictors $2 --seconds 10 --warm 3 2>/dev/null | tail -1 > gpurun_out/r03_zz_ceil_$1_$2.json
  python - gpurun_out/r03_zz_ceil_$1_$2.json "$cfg" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print("native agents, predictors", sys.argv[2], "| pps", d["predictions_per_sec"], "tps", d["train_steps_per_sec"], "batch", d["mean_predict_batch"], "| cpu", d["cgroup"])
PY
done
