#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 120 ./tools/launch_cost > gpurun_out/r03_launch_cost.txt 2>&1; cat gpurun_out/r03_launch_cost.txt
