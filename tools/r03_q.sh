#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 60 python __graft_entry__.py smoke 2>&1 | tail -2
timeout -k 10 260 python tools/e2e_probe.py --agents 32 --predictors 2 --trainers 2 --dynamic --seconds 150 --warm 20 > gpurun_out/r03_q_soak.json 2> gpurun_out/r03_q_soak.err; echo "soak rc=$?"
tail -c 1500 gpurun_out/r03_q_soak.json; grep -i -E "error|traceback|died|failed" gpurun_out/r03_q_soak.err | head -5
