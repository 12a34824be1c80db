#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r5d; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_ops.py -x -q -k "winograd4 or wino4" > $O/pytest_wino4.log 2>&1; echo "pytest rc=$?" | tee $O/rc.txt
tail -3 $O/pytest_wino4.log
timeout 300 python tools/bench_wino.py --iters 10 --wino4 --only 0 1 2 3 4 5 2>&1 | grep -v amdgpu > $O/bench_wino4.txt
cat $O/bench_wino4.txt
export DAWN_HIP_LIB=$PWD/tools/ubench/libdawn_hip_wino4timing.bin
DAWN_WINO4_ABL=64 timeout 200 python tools/bench_wino.py --stamps4 --only 0 2>&1 | grep -v amdgpu > $O/stamps4.txt
for a in 1 4; do
  echo "== DAWN_WINO4_ABL=$a" >> $O/ablations.txt
  DAWN_WINO4_ABL=$a timeout 200 python tools/bench_wino.py --iters 10 --wino4 --only 0 1 2>&1 | grep "F(4x4)" >> $O/ablations.txt
done
head -20 $O/stamps4.txt; cat $O/ablations.txt
