#!/bin/bash
# Round 5, call L: s_memtime stamps + ablations of the FINAL F(4x4) kernel at its shipped shape (level 0, 64 -> 64 channels).
cd "$(dirname "$0")/.."
O=gpurun_out/r5l; mkdir -p $O
timeout 300 python tools/bench_wino.py --iters 10 --wino4 --only 0 1 2>&1 | grep -v amdgpu > $O/bench_wino4.txt
export DAWN_HIP_LIB=$PWD/tools/ubench/libdawn_hip_wino4timing.bin
DAWN_WINO4_ABL=64 timeout 200 python tools/bench_wino.py --stamps4 --only 0 2>&1 | grep -v amdgpu > $O/stamps4.txt
for a in 0 1 2 4 6 7; do
  echo "== DAWN_WINO4_ABL=$a" >> $O/ablations.txt
  DAWN_WINO4_ABL=$a timeout 200 python tools/bench_wino.py --iters 10 --wino4 --only 0 1 2>&1 | grep "F(4x4)" >> $O/ablations.txt
done
cat $O/bench_wino4.txt; head -34 $O/stamps4.txt; cat $O/ablations.txt
