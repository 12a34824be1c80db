#!/bin/bash
# Round 5, call P: what the patch DMA costs the F(2x2) kernel (ablation 32 = no patch DMA in the main loop, 1 = no epilogue, 33 = both) and its phase timeline at K = 1152.
cd "$(dirname "$0")/.."
O=gpurun_out/r5p; mkdir -p $O
export DAWN_HIP_LIB=$PWD/tools/ubench/libdawn_hip_winotiming.bin
for a in 0 32 1 33 0 32; do
  echo "== DAWN_WINO_ABL=$a" >> $O/ablations.txt
  DAWN_WINO_ABL=$a timeout 200 python tools/bench_wino.py --iters 10 --wino-only --only 0 1 3 4 7 11 2>&1 | grep -v amdgpu >> $O/ablations.txt
done
DAWN_WINO_ABL=64 timeout 200 python tools/bench_wino.py --stamps --only 1 2>&1 | grep -v amdgpu > $O/stamps.txt
cat $O/ablations.txt; cat $O/stamps.txt
