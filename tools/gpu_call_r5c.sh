#!/bin/bash
# Round 5, GPU call C: s_memtime timeline and ablations of the F(4x4) kernel (instrumented library).
cd "$(dirname "$0")/.."
O=gpurun_out/r5c; mkdir -p $O
export DAWN_HIP_LIB=$PWD/tools/ubench/libdawn_hip_wino4timing.bin
DAWN_WINO4_ABL=64 timeout 200 python tools/bench_wino.py --stamps4 --only 0 1 > $O/stamps4.txt 2>&1
for a in 0 1 2 4 6 7; do
  echo "== DAWN_WINO4_ABL=$a" >> $O/ablations.txt
  DAWN_WINO4_ABL=$a timeout 200 python tools/bench_wino.py --iters 10 --wino4 --only 0 1 3 2>&1 | grep -v amdgpu >> $O/ablations.txt
done
cat $O/stamps4.txt | head -70; cat $O/ablations.txt
