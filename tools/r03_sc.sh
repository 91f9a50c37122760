#!/bin/bash
# round 3: the state cache -- unit test, the engine tests with it on, engine A/B
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_engine_e2e.py -m gpu -x -q -k "state_cache or gather" > gpurun_out/r03_sc_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -15 gpurun_out/r03_sc_tests.log
[ $rc -eq 0 ] || exit 1
{
for round in 1 2; do for v in 0 1; do
  echo "== engine, 64 agents, --state-cache $v"
  GA3C_TIME_PREDICTIONS=1 timeout -k 10 120 python tools/e2e_probe.py --agents 64 --seconds 8 --warm 3 --state-cache $v 2> gpurun_out/r03_sc_probe_$v.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d.get(k) for k in ('state_cache','predictions_per_sec','train_steps_per_sec','mean_predict_batch','predictor_us_per_batch','agent_wall_us_per_step','agent_cpu_us_per_step','threads_died')}, d['engine']['predict_us_per_call'], d['engine']['train_us_per_call'])"
done; done
} > gpurun_out/r03_sc.txt 2>&1
cat gpurun_out/r03_sc.txt; tail -5 gpurun_out/r03_sc_probe_1.err
