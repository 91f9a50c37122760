#!/bin/bash
# One GPU call that says whether a build is good: the whole -m gpu suite, smoke(), and the driver's bench command line.
#   usage (through gpurun, from the repo root):  bash tools/round_check.sh [tag]
set -o pipefail
TAG=${1:-check}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_tests.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 gpurun_out/${TAG}_tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 90 python __graft_entry__.py smoke 2>&1 | tail -1 || exit 1
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_B128_k20.json 2> gpurun_out/${TAG}_bench_k20.err; echo "bench rc=$?"
python - "$TAG" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/%s_bench_B128_k20.json" % sys.argv[1]).read().strip().splitlines()[-1])
print("value %.3e (wall) / %.3e (GPU span), ms/step %.5f" % (d["value"], d["value_gpu_span"], d["ms_per_step"]),
      {a: round(b / 1e6, 2) for a, b in d["predict_lanes"].items() if a in "1234"})
t = d["train"]
print("train ms/step", round(t["ms_per_step"], 5), "132 rows", round(t["train_132"]["ms_per_step"], 5), "dp 1 rank",
      {k: round(v["ms_per_step"], 5) for k, v in t["train_dp_1rank"].items() if k.startswith("rows_")},
      "roofline", round(d["roofline"]["frac"], 4), d["roofline"].get("profile_frac"))
e = d["e2e"]
print("e2e", round(e["predictions_per_sec"]), round(e["training_steps_per_sec"]), "x2", round(e["agents_x2"]["predictions_per_sec"]),
      round(e["agents_x2"]["training_steps_per_sec"]), "x2 device queue", round(e["agents_x2_frame_queue_on_device"]["predictions_per_sec"]),
      round(e["agents_x2_frame_queue_on_device"]["training_steps_per_sec"]), "| cpu", round(d["cpu_baseline"]["value"]), "placed:", d["cpu_placement"]["why"] if d.get("cpu_placement") else None)
PY
