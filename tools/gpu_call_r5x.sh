#!/bin/bash
# Round 5, call X: h1 written over c1 by the cross-attention epilogues (both hosts): GPU suite, then the whole benchmark against the previous commit's tree (ab_old), alternating.
cd "$(dirname "$0")/.."
O=gpurun_out/r5x; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_full.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $O/pytest_full.log | tail -3
rm -f gpurun_out/ab_bench.txt
timeout 900 bash tools/ab_bench.sh run --steps 2 --warmup 1 --no-shard-sim > $O/ab_bench.log 2>&1; cp gpurun_out/ab_bench.txt $O/; cat $O/ab_bench.txt
