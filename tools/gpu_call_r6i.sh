#!/bin/bash
# Round 6, call I: pixel columns per workgroup of the window-tiled temporal layer (next pixel's rows prefetched under the epilogue): A/B
cd "$(dirname "$0")/.."
O=gpurun_out/r6i; mkdir -p $O
for rep in 1 2; do
for v in 1 2 4 16; do
  echo "== pixels per workgroup $v" | tee -a $O/ppw.txt
  DAWN_TL16_PPW=$v timeout 200 python tools/bench_temporal_layer.py 2>&1 | grep "wmode4" | tail -3 | tee -a $O/ppw.txt
done
done
