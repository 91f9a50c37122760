#!/bin/bash
# round 3, final build: smoke, a 150-s soak with the dynamic adjustment walking NP / NT, then the same with the frame queue on the device
mkdir -p gpurun_out
timeout -k 10 60 python __graft_entry__.py smoke 2>&1 | tail -2
timeout -k 10 260 python tools/e2e_probe.py --agents 48 --predictors 2 --trainers 2 --dynamic --seconds 170 --warm 20 > gpurun_out/r03_q3_soak.json 2> gpurun_out/r03_q3_soak.err; echo "soak rc=$?"
tail -c 1200 gpurun_out/r03_q3_soak.json; grep -i -E "error|traceback|died|failed" gpurun_out/r03_q3_soak.err | head -5
timeout -k 10 160 python tools/e2e_probe.py --agents 48 --predictors 2 --trainers 2 --dynamic --frames planes-device --seconds 60 --warm 10 > gpurun_out/r03_q3_soak_device.json 2> gpurun_out/r03_q3_soak_device.err; echo "soak(device) rc=$?"
tail -c 700 gpurun_out/r03_q3_soak_device.json; grep -i -E "error|traceback|died|failed" gpurun_out/r03_q3_soak_device.err | head -5
