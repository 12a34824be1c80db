#!/bin/bash
# Round 6, call J: variants of the 13-wave temporal layer, alternating on one box (libraries tools/ubench/libdawn_hip_tl13_<variant>.bin)
cd "$(dirname "$0")/.."
O=gpurun_out/r6j; mkdir -p $O
for rep in 1 2 3; do
for v in $VARIANTS; do
  export DAWN_HIP_LIB=$PWD/tools/ubench/libdawn_hip_tl13_$v.bin
  echo "== variant $v" | tee -a $O/variants.txt
  timeout 200 python tools/bench_temporal_layer.py 2>&1 | grep "wmode5" | tail -3 | tee -a $O/variants.txt
done
done
