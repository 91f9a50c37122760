#!/bin/bash
# round 3, final build: the whole -m gpu suite, the driver's bench command line and a long-K run, the profile passes
set -o pipefail
mkdir -p gpurun_out
COMMIT=$1
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r03_y3_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r03_y3_tests.log; tail -4 gpurun_out/r03_y3_tests.log
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r03_y3_bench_B128_k20.json 2> gpurun_out/r03_y3_bench_k20.err && \
timeout -k 10 400 python bench.py --steps 300 --warmup 30 --cpu-seconds 0 --e2e-seconds 0 > gpurun_out/r03_y3_bench_B128_k300.json 2> gpurun_out/r03_y3_bench_k300.err
python - <<'PY'
import json
for k in ("k20","k300"):
    try:
        d=json.loads(open("gpurun_out/r03_y3_bench_B128_%s.json"%k).read().strip().splitlines()[-1])
        print(k, "value %.3e ms/step %.5f wall %.5f"%(d["value"], d["ms_per_step"], d.get("ms_per_step_wall",0)), {a:round(b/1e6,2) for a,b in d["predict_lanes"].items() if a in "1234"}, "train", round(d["train"]["ms_per_step"],5), round(d["train"]["train_132"]["ms_per_step"],5), "roofline", round(d["roofline"]["frac"],4), d["roofline"]["avg_launch_us"])
        if "e2e" in d:
            e=d["e2e"]; print("   e2e", round(e["predictions_per_sec"]), round(e["training_steps_per_sec"]), "x2", round(e["agents_x2"]["predictions_per_sec"]), round(e["agents_x2"]["training_steps_per_sec"]), "dev", round(e["agents_x2_frame_queue_on_device"]["predictions_per_sec"]), round(e["agents_x2_frame_queue_on_device"]["training_steps_per_sec"]), "8q", {a:round(b/1e6,2) for a,b in d.get("predict_lanes_8_hw_queues",{}).items() if a in "1234"})
    except Exception as e: print(k, "failed", e)
PY
bash tools/profile_round.sh r03_y3 $COMMIT > gpurun_out/r03_y3_profile.log 2>&1; tail -2 gpurun_out/r03_y3_profile.log
grep "conv_stack_fwd_kernel<false, false>\|conv_bwd_kernel\|dense1_bwd_tile\|slab_reduce" gpurun_out/r03_y3_pmc_table.md
