#!/bin/bash
# Round 6, call L: split 1x1 tile GEMM with the operand split interleaved between the MFMAs (one basic block per stage) against the
# previous build (tools/ubench/libdawn_hip_gemm_old.bin): tests, microbenchmark of the deep-level shapes, benchmark alternating.
cd "$(dirname "$0")/.."
O=gpurun_out/r6l; mkdir -p $O
(timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_end2end.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3) | tee $O/pytest.log
for v in new old new old; do
  if [ $v = old ]; then export DAWN_HIP_LIB=$PWD/tools/ubench/libdawn_hip_gemm_old.bin; else unset DAWN_HIP_LIB; fi
  echo "== $v" | tee -a $O/ubench.txt
  timeout 300 python tools/bench_gemm1x1_deep.py 2>&1 | grep "M=" | tee -a $O/ubench.txt
done
for rep in 1 2; do
for v in new old; do
  if [ $v = old ]; then export DAWN_HIP_LIB=$PWD/tools/ubench/libdawn_hip_gemm_old.bin; else unset DAWN_HIP_LIB; fi
  timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-decode --no-max-clip --no-shard-sim --no-other-configs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])" | tee -a $O/bench_ab.txt
done
done
