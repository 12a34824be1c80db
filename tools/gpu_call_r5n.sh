#!/bin/bash
# Round 5, call N (a record: the DAWN_WINO_REVERSE knob existed in this call's tree only -- ops.HipOps read it, unet_forward._resblock OR-ed policy bit
# 0x20000000 into conv1 (bit 0) / conv2 (bit 1); the bit is in the shipped default since): Winograd tiles back to front:
# parity, then the whole benchmark alternating over the four settings on ONE box.
cd "$(dirname "$0")/.."
O=gpurun_out/r5n; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_ops.py -x -q -k "reverse or winograd" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/rc.txt
tail -3 $O/pytest.log
for round in 1 2; do
  for r in 0 3 2 1; do
    v=$(DAWN_WINO_REVERSE=$r timeout 300 python bench.py --no-cpu-baseline --no-max-clip --no-decode --no-kernel-events --no-shard-sim --no-other-configs --steps 2 --warmup 1 2>/dev/null | tail -1 |
        python -c "import sys, json; d = json.loads(sys.stdin.read()); print(round(d['value'], 2), round(d['ms_per_step'], 1))")
    echo "round $round DAWN_WINO_REVERSE=$r: $v" | tee -a $O/ab_reverse.txt
  done
done
