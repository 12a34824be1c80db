#!/bin/bash
# Round 6, call K: 1x1 split GEMM with the A-row prefetch a stage earlier (vmcnt leaves it in flight) against the previous build
# (tools/ubench/libdawn_hip_gemm_old.bin): conv tests, in-situ per-shape times of both, benchmark alternating.
cd "$(dirname "$0")/.."
O=gpurun_out/r6k; mkdir -p $O
(timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_end2end.py -m gpu -x -q -k "conv or gemm or end2end or sample" 2>&1 | tail -3) | tee $O/pytest.log
for v in new old; do
  if [ $v = old ]; then export DAWN_HIP_LIB=$PWD/tools/ubench/libdawn_hip_gemm_old.bin; else unset DAWN_HIP_LIB; fi
  timeout 300 python tools/profile_conv_shapes.py 2>&1 | grep -v amdgpu > $O/insitu_shapes_$v.txt
  head -2 $O/insitu_shapes_$v.txt; grep "k=1x1" $O/insitu_shapes_$v.txt | grep -E "M=12800|M=51200|N=768" | head -30
done
for rep in 1 2; do
for v in new old; do
  if [ $v = old ]; then export DAWN_HIP_LIB=$PWD/tools/ubench/libdawn_hip_gemm_old.bin; else unset DAWN_HIP_LIB; fi
  timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-decode --no-max-clip --no-shard-sim --no-other-configs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])" | tee -a $O/bench_ab.txt
done
done
