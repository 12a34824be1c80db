#!/bin/bash
# Round 5, call Z: resampling convs with the next K block's rows in flight under the current block's multiplies: parity, isolated A/B against the previous build
# (alternating), whole benchmark A/B.
cd "$(dirname "$0")/.."
O=gpurun_out/r5z; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_ops.py -x -q -k "resample or transposed or conv_gemm or gemm1x1" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
for rep in 1 2; do
  for which in prev new; do
    if [ $which = prev ]; then export DAWN_HIP_LIB=$PWD/tools/ubench/libdawn_hip_prev.bin; else unset DAWN_HIP_LIB; fi
    echo "== $which (pass $rep)" >> $O/ab_isolated.txt
    timeout 300 python tools/bench_resample.py --policies 0 2>&1 | grep -v amdgpu >> $O/ab_isolated.txt
  done
done
cat $O/ab_isolated.txt
for round in 1 2 3; do
  for which in prev new; do
    if [ $which = prev ]; then export DAWN_HIP_LIB=$PWD/tools/ubench/libdawn_hip_prev.bin; else unset DAWN_HIP_LIB; fi
    v=$(timeout 300 python bench.py --no-cpu-baseline --no-max-clip --no-decode --no-kernel-events --no-shard-sim --no-other-configs --steps 2 --warmup 1 2>/dev/null | tail -1 |
        python -c "import sys, json; d = json.loads(sys.stdin.read()); print(round(d['value'], 2), round(d['ms_per_step'], 1))")
    echo "round $round $which: $v" | tee -a $O/ab_bench.txt
  done
done
unset DAWN_HIP_LIB
