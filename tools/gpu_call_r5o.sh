#!/bin/bash
# Round 5, call O: non-temporal output stores in the Winograd kernels (-DDAWN_WINO_ST_AUX=2 build in tools/ubench/libdawn_hip_nt.bin) against the
# shipped build: isolated per shape and inside the whole benchmark, alternating on ONE box.
cd "$(dirname "$0")/.."
O=gpurun_out/r5o; mkdir -p $O
for rep in 1 2; do
  for which in base nt; do
    if [ $which = nt ]; then export DAWN_HIP_LIB=$PWD/tools/ubench/libdawn_hip_nt.bin; else unset DAWN_HIP_LIB; fi
    echo "== $which (pass $rep)" >> $O/ab_isolated.txt
    timeout 300 python tools/bench_wino.py --iters 10 --wino4 --wino-only --only 0 1 2 3 4 5 6 7 2>&1 | grep -v amdgpu >> $O/ab_isolated.txt
  done
done
for round in 1 2 3; do
  for which in base nt; do
    if [ $which = nt ]; then export DAWN_HIP_LIB=$PWD/tools/ubench/libdawn_hip_nt.bin; else unset DAWN_HIP_LIB; fi
    v=$(timeout 300 python bench.py --no-cpu-baseline --no-max-clip --no-decode --no-kernel-events --no-shard-sim --no-other-configs --steps 2 --warmup 1 2>/dev/null | tail -1 |
        python -c "import sys, json; d = json.loads(sys.stdin.read()); print(round(d['value'], 2), round(d['ms_per_step'], 1))")
    echo "round $round $which: $v" | tee -a $O/ab_bench.txt
  done
done
unset DAWN_HIP_LIB
