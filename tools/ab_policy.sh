#!/bin/bash
# GPU box: A/B of conv kernel policies inside the whole benchmark, alternating within ONE gpurun call (boxes and power /
# thermal state differ between calls).   bash tools/ab_policy.sh "0x580D 0x5C0D 0x5C4D" [rounds] [extra bench.py args]
cd "$(dirname "$0")/.."
POL=${1:-"0x580D 0x5C0D"}; R=${2:-2}; shift; shift
mkdir -p gpurun_out
for round in $(seq 1 $R); do
  for p in $POL; do
    v=$(timeout 300 python bench.py --no-cpu-baseline --no-max-clip --no-decode --no-kernel-events --no-shard-sim --no-other-configs --steps 2 --warmup 1 --conv-policy $p "$@" 2>/dev/null | tail -1 |
        python -c "import sys, json; d = json.loads(sys.stdin.read()); print(round(d['value'], 2), round(d['ms_per_step'], 1))")
    echo "round $round policy $p: $v" | tee -a gpurun_out/ab_policy.txt
  done
done
