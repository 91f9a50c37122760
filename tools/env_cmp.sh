#!/bin/bash
# Development aid: bench.py legs under two settings of one environment switch, interleaved.
#   usage (through gpurun): bash tools/env_cmp.sh GA3C_D1F_TILE 1 0 [extra bench args]
set -e
VAR=$1; A=$2; B=$3; shift 3
for round in 1 2; do
  for v in $A $B; do
    env $VAR=$v timeout -k 10 200 python bench.py --steps 300 --warmup 30 --cpu-seconds 0 --e2e-seconds 0 "$@" > gpurun_out/envcmp.json
    python - "$VAR=$v" <<'PY'
import json, sys
d = json.loads(open('gpurun_out/envcmp.json').read().strip().splitlines()[-1])
pl = d['predict_lanes']
print(sys.argv[1], 'lanes 1/2/3: %.2f %.2f %.2f M/s' % (pl['1'] / 1e6, pl['2'] / 1e6, pl['3'] / 1e6), 'train %.0f' % d['train']['value'],
      'u8 %.2f M/s' % (d['uint8_resident']['predictions_per_sec'] / 1e6))
PY
  done
done
