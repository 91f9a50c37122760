#!/bin/bash
# round 3, call v: does any HIP runtime switch shorten the kernel-to-kernel boundary?  (lanes 1-2 prediction, 1 train lane)
set -e
mkdir -p gpurun_out
out=gpurun_out/r03_v_runtime_knobs.txt
: > $out
for cfg in "X=0" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "DEBUG_HIP_KERNARG_COPY_OPT=0" "ROC_SKIP_KERNEL_ARG_COPY=1" "ROC_USE_FGS_KERNARG=0" \
           "AMD_OPT_FLUSH=0" "ROC_SYSTEM_SCOPE_SIGNAL=0" "AMD_DIRECT_DISPATCH=0" "DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1" "GPU_FLUSH_ON_EXECUTION=1" \
           "ROC_ACTIVE_WAIT_TIMEOUT=0" "DEBUG_HIP_DYNAMIC_QUEUES=0" "GPU_STREAMOPS_CP_WAIT=1" "X=1"; do
  echo "== $cfg" >> $out
  env $cfg timeout -k 10 120 python tools/lanes.py 128 2 >> $out 2>&1 || echo "FAILED" >> $out
  env $cfg timeout -k 10 120 python tools/train_lanes.py 128 1 >> $out 2>&1 || echo "FAILED" >> $out
  echo "progress $cfg"
done
cat $out
