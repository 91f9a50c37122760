#!/bin/bash
# round 3, first GPU call: the whole -m gpu suite, K-invariance of the headline, trainer pipelining, engine regimes
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r03_a_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r03_a_tests.log
tail -5 gpurun_out/r03_a_tests.log
timeout -k 10 120 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --e2e-seconds 0 > gpurun_out/r03_a_bench_k20.json 2> gpurun_out/r03_a_bench_k20.err && \
timeout -k 10 120 python bench.py --steps 300 --warmup 30 --cpu-seconds 0 --e2e-seconds 0 > gpurun_out/r03_a_bench_k300.json 2> gpurun_out/r03_a_bench_k300.err
python - <<'PY'
import json
for k in ("k20","k300"):
    try:
        d=json.loads(open("gpurun_out/r03_a_bench_%s.json"%k).read().strip().splitlines()[-1])
        print(k, "value %.3e ms/step %.5f inner %.5f"%(d["value"], d["ms_per_step"], d.get("ms_per_step_wall",0)), d["predict_lanes"], "train", d["train"]["ms_per_step"], d["train"].get("train_132"), d.get("predict_lanes_8_hw_queues"))
    except Exception as e: print(k, "failed", e)
PY
timeout -k 10 200 python tools/train_latency.py 128 132 > gpurun_out/r03_a_train_latency.txt 2>&1; cat gpurun_out/r03_a_train_latency.txt | tail -8
for cfg in "2 2 4" "2 4 4" "4 4 4" "4 4 8" "2 4 8" "2 2 8"; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$3 timeout -k 10 90 python tools/e2e_probe.py --agents 64 --predictors $1 --lanes $2 --seconds 10 --warm 4 2>/dev/null | tail -1 > gpurun_out/r03_a_probe_p$1_l$2_q$3.json
  echo "pred $1 lanes $2 queues $3: $(cat gpurun_out/r03_a_probe_p$1_l$2_q$3.json)"
done
