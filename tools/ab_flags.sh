#!/bin/bash
# GPU box: whole-benchmark A/B of bench.py argument sets, alternating inside ONE gpurun call.
#   bash tools/ab_flags.sh 3 "" "--temporal-attn-flags 1"
cd "$(dirname "$0")/.."
R=$1; shift
mkdir -p gpurun_out
for round in $(seq 1 $R); do
  for f in "$@"; do
    v=$(timeout 300 python bench.py --no-cpu-baseline --no-max-clip --no-decode --no-kernel-events --no-shard-sim --no-other-configs --steps 2 --warmup 1 $f 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(round(d['value'], 2), round(d['ms_per_step'], 1))")
    echo "round $round [$f]: $v" | tee -a gpurun_out/ab_flags.txt
  done
done
