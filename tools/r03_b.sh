#!/bin/bash
# round 3, second GPU call: queue map, K-invariance with GPU-span timing, trainer pipelining with the narrow gather, engine regimes
set -o pipefail
mkdir -p gpurun_out
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -o tools/qmap tools/qmap.hip 2>/dev/null
for q in 4 8; do GPU_MAX_HW_QUEUES=$q timeout -k 10 120 ./tools/qmap 8 > gpurun_out/r03_b_qmap_q$q.txt 2>&1; cat gpurun_out/r03_b_qmap_q$q.txt; done
timeout -k 10 300 python -m pytest tests/test_gpu_train_parity.py tests/test_gpu_engine_e2e.py tests/test_gpu_frontend.py -m gpu -x -q > gpurun_out/r03_b_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03_b_tests.log
timeout -k 10 120 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --e2e-seconds 0 > gpurun_out/r03_b_bench_k20.json 2> gpurun_out/r03_b_bench_k20.err && \
timeout -k 10 120 python bench.py --steps 300 --warmup 30 --cpu-seconds 0 --e2e-seconds 0 > gpurun_out/r03_b_bench_k300.json 2> gpurun_out/r03_b_bench_k300.err
python - <<'PY'
import json
for k in ("k20","k300"):
    try:
        d=json.loads(open("gpurun_out/r03_b_bench_%s.json"%k).read().strip().splitlines()[-1])
        print(k, "value %.3e ms/step %.5f wall %.5f"%(d["value"], d["ms_per_step"], d.get("ms_per_step_wall",0)), {a:round(b/1e6,2) for a,b in d["predict_lanes"].items() if a in "1234"}, "train", d["train"]["ms_per_step"], d["train"].get("train_132",{}).get("ms_per_step"), {a:round(b/1e6,2) for a,b in d.get("predict_lanes_8_hw_queues",{}).items() if a in "1234"})
    except Exception as e: print(k, "failed", e)
PY
timeout -k 10 200 python tools/train_latency.py 128 132 > gpurun_out/r03_b_train_latency.txt 2>&1; tail -6 gpurun_out/r03_b_train_latency.txt
for cfg in "2 2 4" "3 3 4" "4 4 4" "4 4 8"; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$3 timeout -k 10 90 python tools/e2e_probe.py --agents 64 --predictors $1 --lanes $2 --seconds 10 --warm 4 2>/dev/null | tail -1 > gpurun_out/r03_b_probe_p$1_l$2_q$3.json
  python - gpurun_out/r03_b_probe_p$1_l$2_q$3.json "$cfg" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print("pred lanes queues", sys.argv[2], "| pps", d["predictions_per_sec"], "tps", d["train_steps_per_sec"], "batch", d["mean_predict_batch"], "| predict", d["engine"]["predict_us_per_call"], "| train", d["engine"]["train_us_per_call"], "reader waits", d["engine"]["train_reader_waits_per_call"], "| cpu", d["cgroup"])
PY
done
