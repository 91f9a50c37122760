#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -o tools/qmap tools/qmap.hip 2>/dev/null
GPU_MAX_HW_QUEUES=4 timeout -k 10 120 ./tools/qmap 8 null > gpurun_out/r03_c_qmap_q4_null.txt 2>&1; head -16 gpurun_out/r03_c_qmap_q4_null.txt
timeout -k 10 200 python tools/train_latency.py 128 132 > gpurun_out/r03_c_train_latency.txt 2>&1; tail -6 gpurun_out/r03_c_train_latency.txt
GA3C_TRAIN_PRIORITY=0 timeout -k 10 200 python tools/train_latency.py 128 132 > gpurun_out/r03_c_train_latency_plain.txt 2>&1; tail -6 gpurun_out/r03_c_train_latency_plain.txt
for B in 128 132; do
timeout -k 10 120 python tools/ktime.py --batch $B @predict @train conv1_fwd conv2_fwd dense1_fwd heads dense1_bwd_tile conv2_dw conv2_dx conv1_dw slab_reduce > gpurun_out/r03_c_ktime_$B.txt 2>&1; echo "B=$B"; cat gpurun_out/r03_c_ktime_$B.txt
done
timeout -k 10 120 python tools/ktime.py --batch 128 conv_stack_fwd conv_stack_fwd_train conv_bwd > gpurun_out/r03_c_ktime_128b.txt 2>&1; cat gpurun_out/r03_c_ktime_128b.txt
for cfg in "2 2 4" "4 4 4"; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$3 timeout -k 10 90 python tools/e2e_probe.py --agents 64 --predictors $1 --lanes $2 --seconds 10 --warm 4 2>/dev/null | tail -1 > gpurun_out/r03_c_probe_p$1_l$2_q$3.json
  python - gpurun_out/r03_c_probe_p$1_l$2_q$3.json "$cfg" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print("pred lanes queues", sys.argv[2], "| pps", d["predictions_per_sec"], "tps", d["train_steps_per_sec"], "batch", d["mean_predict_batch"], "| predict", d["engine"]["predict_us_per_call"], "| train", d["engine"]["train_us_per_call"], "reader waits", d["engine"]["train_reader_waits_per_call"], "| cpu", d["cgroup"])
PY
done
