#!/bin/bash
# Round 6, call O: row-stationary GEMM with bias / residual / tr applied behind the staging tile (coalesced loads) against the previous build
cd "$(dirname "$0")/.."
O=gpurun_out/r6o; mkdir -p $O
(timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_end2end.py -m gpu -x -q -k "conv or gemm or end2end or sample or unet or forward" 2>&1 | grep -E "passed|failed|error" | tail -3) | tee $O/pytest.log
for v in new old new old; do
  if [ $v = old ]; then export DAWN_HIP_LIB=$PWD/tools/ubench/libdawn_hip_rowreg_old.bin; else unset DAWN_HIP_LIB; fi
  echo "== $v" | tee -a $O/ubench.txt
  for i in 9 3 4 8; do timeout 100 python tools/bench_gemm1x1.py --only $i 2>&1 | grep "policy" | tee -a $O/ubench.txt; done
done
for rep in 1 2; do
for v in new old; do
  if [ $v = old ]; then export DAWN_HIP_LIB=$PWD/tools/ubench/libdawn_hip_rowreg_old.bin; else unset DAWN_HIP_LIB; fi
  timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-decode --no-max-clip --no-shard-sim --no-other-configs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])" | tee -a $O/bench_ab.txt
done
done
