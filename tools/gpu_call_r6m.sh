#!/bin/bash
# Round 6, call M: SLA apply kernel with the out product on the bf16 pipe against the fp32-pipe build (tools/ubench/libdawn_hip_sla_fp32out.bin)
cd "$(dirname "$0")/.."
O=gpurun_out/r6m; mkdir -p $O
(timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_end2end.py -m gpu -x -q -k "sla or end2end or sample or unet or forward" 2>&1 | grep -E "passed|failed|error" | tail -3) | tee $O/pytest.log
for v in new old new old; do
  if [ $v = old ]; then export DAWN_HIP_LIB=$PWD/tools/ubench/libdawn_hip_sla_fp32out.bin; else unset DAWN_HIP_LIB; fi
  echo -n "$v: " | tee -a $O/ubench.txt; timeout 300 python tools/bench_sla_layer.py 2>&1 | grep sla_layer | tee -a $O/ubench.txt
done
for rep in 1 2; do
for v in new old; do
  if [ $v = old ]; then export DAWN_HIP_LIB=$PWD/tools/ubench/libdawn_hip_sla_fp32out.bin; else unset DAWN_HIP_LIB; fi
  timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-decode --no-max-clip --no-shard-sim --no-other-configs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])" | tee -a $O/bench_ab.txt
done
done
