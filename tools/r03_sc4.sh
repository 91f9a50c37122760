#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for v in 1 0; do echo "== GA3C_FRAMES_IN_LINE=$v"; GA3C_FRAMES_IN_LINE=$v timeout -k 10 200 python tools/train_latency.py 128 132 2>&1 | grep "train_frames"; done
for v in 1 0 1 0; do
  GA3C_FRAMES_IN_LINE=$v timeout -k 10 120 python tools/e2e_probe.py --agents 64 --seconds 8 --warm 3 --frames planes-device 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('in_line=$v', {k:d.get(k) for k in ('predictions_per_sec','train_steps_per_sec','mean_predict_batch','agent_wall_us_per_step','threads_died')}, d['engine']['train_us_per_call'])"
done
