#!/bin/bash
# Round 5, GPU call B: first run of the F(4x4,3x3) kernel: parity tests, then isolated timing against F(2x2) / direct at levels 0-1.
cd "$(dirname "$0")/.."
O=gpurun_out/r5b; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_ops.py -x -q -k "winograd4 or wino4" > $O/pytest_wino4.log 2>&1; echo "pytest rc=$?" | tee $O/rc.txt
tail -30 $O/pytest_wino4.log
timeout 300 python tools/bench_wino.py --iters 10 --wino4 --only 0 1 2 3 4 5 > $O/bench_wino4.txt 2>&1
cat $O/bench_wino4.txt
