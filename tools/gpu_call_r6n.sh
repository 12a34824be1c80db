#!/bin/bash
# Round 6, call N: benchmark with the 13-wave attention core of the unfused levels against the round-5 choice (tools/ubench/libdawn_hip_noattn13.bin)
cd "$(dirname "$0")/.."
O=gpurun_out/r6n; mkdir -p $O
for rep in 1 2 3; do
for v in new old; do
  if [ $v = old ]; then export DAWN_HIP_LIB=$PWD/tools/ubench/libdawn_hip_noattn13.bin; else unset DAWN_HIP_LIB; fi
  timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-decode --no-max-clip --no-shard-sim --no-other-configs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])" | tee -a $O/bench_ab.txt
done
done
