#!/bin/bash
# round 3: the state cache on by default -- train-call latency by name, engine tests, soak
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 200 python tools/train_latency.py 128 132 > gpurun_out/r03_sc2_train_latency.txt 2>&1; echo "train_latency rc=$?"; cat gpurun_out/r03_sc2_train_latency.txt
timeout -k 10 300 python -m pytest tests/test_gpu_engine_e2e.py -m gpu -x -q -k "engine_trains" > gpurun_out/r03_sc2_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03_sc2_tests.log
timeout -k 10 200 python tools/e2e_probe.py --agents 48 --predictors 2 --trainers 2 --dynamic --seconds 100 --warm 15 > gpurun_out/r03_sc2_soak.json 2> gpurun_out/r03_sc2_soak.err; echo "soak rc=$?"
tail -c 900 gpurun_out/r03_sc2_soak.json; grep -i -E "error|traceback|died|failed" gpurun_out/r03_sc2_soak.err | head -5
