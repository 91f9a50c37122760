#!/bin/bash
# One GPU call, many cases of one measuring tool.  A case is "tag|ENV=.. ENV=..|taskset cpu list or -|tool args".
#   tools/run_matrix.sh <tool.py> <out prefix> <case> [<case> ...]        results: gpurun_out/<prefix>_<tag>.json
# The tool prints one JSON line last; a digest of it is echoed per case so that the call's tail shows the whole matrix.
mkdir -p gpurun_out
tool=$1; prefix=$2; shift 2
for case in "$@"; do
  IFS='|' read -r tag envs cpus args <<< "$case"
  out=gpurun_out/${prefix}_${tag}.json
  pre=""
  if [ "$cpus" != "-" ] && [ -n "$cpus" ]; then pre="taskset -c $cpus"; fi
  env $envs timeout -k 10 200 $pre python $tool $args 2>gpurun_out/${prefix}_${tag}.err | tail -1 > $out
  python - "$tag" "$out" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read())
    keys = ('predictions_per_sec', 'train_steps_per_sec', 'training_steps_per_sec', 'mean_predict_batch', 'predictor_us_per_batch',
            'us_cpu_per_prediction', 'answer_to_running_us', 'cgroup')
    pl = d.get('placement') or {}
    print(sys.argv[1], {k: d[k] for k in keys if k in d}, 'cpus', len(pl.get('cpus', [])) or 'unplaced', flush=True)
except Exception as e:
    print(sys.argv[1], "FAILED", e, flush=True)
PY
done
