#!/bin/bash
# round 3, last build: an 8-minute soak, 64 agents, ThreadDynamicAdjustment walking NT / NP / NA every 2 s, rollouts naming their states
mkdir -p gpurun_out
timeout -k 10 620 python tools/e2e_probe.py --agents 64 --predictors 2 --trainers 2 --dynamic --seconds 480 --warm 30 > gpurun_out/r03_q4_soak.json 2> gpurun_out/r03_q4_soak.err &
pid=$!
while kill -0 $pid 2>/dev/null; do sleep 60; echo "soak running $(date +%T)"; done
wait $pid; echo "soak rc=$?"
tail -c 1300 gpurun_out/r03_q4_soak.json; grep -i -E "error|traceback|died|failed" gpurun_out/r03_q4_soak.err | head -5
