#!/bin/bash
# Round 4, the split path at the rows a "batch = 128" trainer assembles (129 .. 133): conv2_dw cut for three workgroups per
# CU, dense1_bwd_tile's tail rows beside the first chunk, dense1/w stepped beside conv2_dx, conv2_dw + conv1_dw in one launch, wide uint8 staging.  Per-kernel times, whole steps
# with the switches on / off, the train and gradient parity tests.
#   usage (through gpurun, from the repo root):  bash tools/r04_cliff_ab.sh
set -o pipefail
mkdir -p gpurun_out
O=gpurun_out/r04_cliff_ab.txt
: > $O
for B in 132 129 140 192; do
  echo "## B = $B (defaults: GA3C_C2DW_OCC=3 GA3C_D1B_TAIL=1 GA3C_WD_STEP_IN_CONV2_DX=1 GA3C_DW_PAIR=1, conv1_dw workgroups by conv1_dw_blocks())" >> $O
  timeout -k 10 150 python tools/ktime.py --batch $B conv2_dw conv2_dw_occ3 conv1_dw conv_dw_pair dense1_bwd_tile_notail dense1_bwd_tile @train >> $O 2>&1 || exit 1
  echo "## ... uint8 states" >> $O
  timeout -k 10 150 python tools/ktime.py --u8 --batch $B conv1_fwd_u8 conv1_dw_u8 @train >> $O 2>&1 || exit 1
  echo "## B = $B, GA3C_C2DW_OCC=2 GA3C_D1B_TAIL=0 GA3C_WD_STEP_IN_CONV2_DX=0 GA3C_DW_PAIR=0 GA3C_C1DW_BLOCKS=512 (the launches of the build before, f32 / uint8)" >> $O
  GA3C_C2DW_OCC=2 GA3C_D1B_TAIL=0 GA3C_WD_STEP_IN_CONV2_DX=0 GA3C_DW_PAIR=0 GA3C_C1DW_BLOCKS=512 timeout -k 10 150 python tools/ktime.py --batch $B @train >> $O 2>&1 || exit 1
  GA3C_C2DW_OCC=2 GA3C_D1B_TAIL=0 GA3C_WD_STEP_IN_CONV2_DX=0 GA3C_DW_PAIR=0 GA3C_C1DW_BLOCKS=512 timeout -k 10 150 python tools/ktime.py --u8 --batch $B @train >> $O 2>&1 || exit 1
done
echo "## B = 128" >> $O
timeout -k 10 150 python tools/ktime.py --batch 128 dense1_bwd_tile @train @predict >> $O 2>&1 || exit 1
echo "## bits (tools/ab_bits.py), defaults" >> $O
timeout -k 10 200 python tools/ab_bits.py 129 132 133 134 > gpurun_out/r04_bits_new.txt 2>&1 || { cat gpurun_out/r04_bits_new.txt; exit 1; }
GA3C_C2DW_OCC=2 GA3C_D1B_TAIL=0 GA3C_WD_STEP_IN_CONV2_DX=0 GA3C_DW_PAIR=0 timeout -k 10 200 python tools/ab_bits.py 129 132 133 134 > gpurun_out/r04_bits_old.txt 2>&1 || { cat gpurun_out/r04_bits_old.txt; exit 1; }
cat gpurun_out/r04_bits_new.txt >> $O
if cmp -s gpurun_out/r04_bits_new.txt gpurun_out/r04_bits_old.txt; then echo "bits: identical with the switches off" >> $O; else echo "bits: DIFFER with the switches off" >> $O; fi
cat $O
timeout -k 10 600 python -m pytest tests/test_gpu_train_parity.py tests/test_gpu_parity.py -x -q > gpurun_out/r04_cliff_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r04_cliff_tests.log
exit $rc
