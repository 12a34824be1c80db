#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r5h; mkdir -p $O
timeout 300 python tools/bench_wino.py --iters 10 --wino4 --stagger 3 4 5 9 10 11 --only 0 2>&1 | grep -v amdgpu | tee $O/bench_wino4_stagger.txt
