#!/bin/bash
set -e
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/dh_par.log 2>&1 || { tail -30 gpurun_out/dh_par.log; exit 1; }
tail -2 gpurun_out/dh_par.log
timeout -k 10 120 python tools/ktime.py dense1_fwd heads dense1_heads
for v in 1 0 1 0; do
  echo D1_HEADS=$v
  GA3C_D1_HEADS=$v timeout -k 10 200 python bench.py --steps 300 --warmup 30 --cpu-seconds 0 --e2e-seconds 0 > gpurun_out/dh_bench_$v.json
  python - $v <<'PY'
import json, sys
d = json.loads(open('gpurun_out/dh_bench_%s.json' % sys.argv[1]).read().strip().splitlines()[-1])
print(d['predict_lanes'], d['stream_ms_per_step'], d['train']['value'], d['train']['hogwild_lanes'])
PY
done
