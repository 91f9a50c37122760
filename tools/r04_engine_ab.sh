#!/bin/bash
# The engine at its ceiling (256 native agents, frame queue on the device) with the split path's switches on / off, interleaved.
#   usage (through gpurun): bash tools/r04_engine_ab.sh
mkdir -p gpurun_out; O=gpurun_out/r04_engine_ab.txt; : > $O
for round in 1 2 3; do
  echo "## defaults (round $round)" >> $O
  timeout -k 10 120 python tools/engine_ceiling.py --agents 256 --frame-queue-on-device --seconds 12 --warm 4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: (round(v, 1) if isinstance(v, float) else v) for k, v in d.items() if isinstance(v, (int, float))})" >> $O 2>&1
  echo "## GA3C_C2DW_OCC=2 GA3C_D1B_TAIL=0 GA3C_WD_STEP_IN_CONV2_DX=0 GA3C_DW_PAIR=0 GA3C_C1DW_BLOCKS=512 (round $round)" >> $O
  GA3C_C2DW_OCC=2 GA3C_D1B_TAIL=0 GA3C_WD_STEP_IN_CONV2_DX=0 GA3C_DW_PAIR=0 GA3C_C1DW_BLOCKS=512 timeout -k 10 120 python tools/engine_ceiling.py --agents 256 --frame-queue-on-device --seconds 12 --warm 4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: (round(v, 1) if isinstance(v, float) else v) for k, v in d.items() if isinstance(v, (int, float))})" >> $O 2>&1
done
cat $O
