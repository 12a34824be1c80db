#!/bin/bash
# Round 6, call G: static wave priorities in the window-tiled temporal layer (A/B: none / younger half / older half)
cd "$(dirname "$0")/.."
O=gpurun_out/r6g; mkdir -p $O
for rep in 1 2; do
for v in none YOUNG OLD; do
  if [ $v = none ]; then unset DAWN_HIP_LIB; else export DAWN_HIP_LIB=$PWD/tools/ubench/libdawn_hip_tl16prio_$v.bin; fi
  echo "== priority $v" | tee -a $O/prio.txt
  timeout 200 python tools/bench_temporal_layer.py 2>&1 | grep "HW=4096 wmode4\|segment.*wmode4" | tail -2 | tee -a $O/prio.txt
done
done
