#!/bin/bash
# Round 5, call M: fewer vector instructions per tile in both Winograd kernels (tile stepping without divisions, outputs through a buffer
# descriptor, first chunk accumulates from the zero operand): parity, then isolated A/B against the previous build (alternating, one box).
cd "$(dirname "$0")/.."
O=gpurun_out/r5m; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -k "wino or conv3x3 or conv_gemm" > $O/pytest_conv.log 2>&1; echo "pytest rc=$?" | tee $O/rc.txt
tail -3 $O/pytest_conv.log
for rep in 1 2; do
  for which in prev new; do
    if [ $which = prev ]; then export DAWN_HIP_LIB=$PWD/tools/ubench/libdawn_hip_prev.bin; else unset DAWN_HIP_LIB; fi
    echo "== $which (pass $rep)" >> $O/ab_isolated.txt
    timeout 300 python tools/bench_wino.py --iters 10 --wino4 --wino-only 2>&1 | grep -v amdgpu >> $O/ab_isolated.txt
  done
done
unset DAWN_HIP_LIB
cat $O/ab_isolated.txt
