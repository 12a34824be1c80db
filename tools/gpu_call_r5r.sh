#!/bin/bash
# Round 5, call R: F(2x2) patch DMA addressed through a frame descriptor + one vector add per slot (11 vector instructions per slot before): parity,
# isolated A/B against the previous build (alternating, one box), whole benchmark A/B.
cd "$(dirname "$0")/.."
O=gpurun_out/r5r; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -k "wino or conv3x3 or conv_gemm" > $O/pytest_conv.log 2>&1; echo "pytest rc=$?" | tee $O/rc.txt
tail -3 $O/pytest_conv.log
for rep in 1 2; do
  for which in prev new; do
    if [ $which = prev ]; then export DAWN_HIP_LIB=$PWD/tools/ubench/libdawn_hip_prev.bin; else unset DAWN_HIP_LIB; fi
    echo "== $which (pass $rep)" >> $O/ab_isolated.txt
    timeout 300 python tools/bench_wino.py --iters 10 --wino-only 2>&1 | grep -v amdgpu >> $O/ab_isolated.txt
  done
done
for round in 1 2 3; do
  for which in prev new; do
    if [ $which = prev ]; then export DAWN_HIP_LIB=$PWD/tools/ubench/libdawn_hip_prev.bin; else unset DAWN_HIP_LIB; fi
    v=$(timeout 300 python bench.py --no-cpu-baseline --no-max-clip --no-decode --no-kernel-events --no-shard-sim --no-other-configs --steps 2 --warmup 1 2>/dev/null | tail -1 |
        python -c "import sys, json; d = json.loads(sys.stdin.read()); print(round(d['value'], 2), round(d['ms_per_step'], 1))")
    echo "round $round $which: $v" | tee -a $O/ab_bench.txt
  done
done
unset DAWN_HIP_LIB
