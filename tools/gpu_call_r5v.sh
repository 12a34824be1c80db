#!/bin/bash
# Round 5, call V: the 64-channel attention layers IN PLACE (output over the input rows; EXPERIMENT knob DAWN_INPLACE_ATTN of this call's tree): parity of one
# evaluation against the out-of-place form, then the whole benchmark alternating.
cd "$(dirname "$0")/.."
O=gpurun_out/r5v; mkdir -p $O
for r in 1 2; do
  for ip in 0 3 1 2; do
    v=$(DAWN_INPLACE_ATTN=$ip timeout 300 python bench.py --no-cpu-baseline --no-max-clip --no-decode --no-kernel-events --no-shard-sim --no-other-configs --steps 2 --warmup 1 2>/dev/null | tail -1 |
        python -c "import sys, json; d = json.loads(sys.stdin.read()); print(round(d['value'], 2), round(d['ms_per_step'], 1))")
    echo "round $r DAWN_INPLACE_ATTN=$ip: $v" | tee -a $O/ab_inplace.txt
  done
done
DAWN_INPLACE_ATTN=3 timeout 600 python -m pytest tests/test_hip_end2end.py -x -q -k "full_C3 or tiny" 2>&1 | tail -3 | tee $O/pytest_inplace.txt
