#!/bin/bash
# Round 6, call C: whole-benchmark A/B of the fused temporal layer: 32 x 32 kernel (--temporal-flags 256) vs window-tiled (default), alternating.
cd "$(dirname "$0")/.."
O=gpurun_out/r6c; mkdir -p $O
for round in 1 2 3; do
  for which in old new; do
    if [ $which = old ]; then TF=256; else TF=0; fi
    v=$(timeout 400 python bench.py --no-cpu-baseline --no-max-clip --no-decode --no-kernel-events --no-shard-sim --no-other-configs --steps 2 --warmup 1 --temporal-flags $TF 2>/dev/null | tail -1 |
        python -c "import sys, json; d = json.loads(sys.stdin.read()); print(round(d['value'], 2), round(d['ms_per_step'], 1))")
    echo "round $round $which (temporal flags $TF): $v" | tee -a $O/ab_bench.txt
  done
done
# configs[1]: 128 x 128, 400 frames
for which in old new; do
  if [ $which = old ]; then TF=256; else TF=0; fi
  v=$(timeout 400 python bench.py --res 128 --frames 400 --no-cpu-baseline --no-max-clip --no-decode --no-kernel-events --no-shard-sim --no-other-configs --steps 2 --warmup 1 --temporal-flags $TF 2>/dev/null | tail -1 |
      python -c "import sys, json; d = json.loads(sys.stdin.read()); print(round(d['value'], 2), round(d['ms_per_step'], 1))")
  echo "configs[1] 128px 400f $which (temporal flags $TF): $v" | tee -a $O/ab_bench.txt
done
