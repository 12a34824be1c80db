#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r5e; mkdir -p $O
export DAWN_HIP_LIB=$PWD/tools/ubench/libdawn_hip_wino4timing.bin
for a in 64 68 66 70; do
  echo "== DAWN_WINO4_ABL=$a" >> $O/stamps4.txt
  DAWN_WINO4_ABL=$a timeout 200 python tools/bench_wino.py --stamps4 --only 0 2>&1 | grep -v amdgpu | grep -A17 "workgroup 5" >> $O/stamps4.txt
done
cat $O/stamps4.txt
