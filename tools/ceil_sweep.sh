#!/bin/bash
# usage (through gpurun): bash tools/ceil_sweep.sh "<agents> <predictors> <blocking 0|1>" ...
for cfg in "$@"; do
  set -- $cfg
  u0=$(grep usage_usec /sys/fs/cgroup/cpu.stat | cut -d' ' -f2); t0=$(grep nr_throttled /sys/fs/cgroup/cpu.stat | cut -d' ' -f2)
  GA3C_BLOCKING_SYNC=$3 timeout -k 10 120 python tools/engine_ceiling.py --agents $1 --predictors $2 --seconds 10 --warm 3 --no-train 2>&1 | tail -1 > gpurun_out/ceil.json
  u1=$(grep usage_usec /sys/fs/cgroup/cpu.stat | cut -d' ' -f2); t1=$(grep nr_throttled /sys/fs/cgroup/cpu.stat | cut -d' ' -f2)
  python - "$cfg" $((u1-u0)) $((t1-t0)) <<'PY'
import json, sys
d = json.loads(open('gpurun_out/ceil.json').read())
print(sys.argv[1], '| pred/s', d['predictions_per_sec'], 'batch', d['mean_predict_batch'], d['predictor_us_per_batch'], '| cpu-s', round(int(sys.argv[2]) / 1e6, 1), 'throttled periods', sys.argv[3])
PY
done
