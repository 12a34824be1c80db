#!/bin/bash
# Round 5, GPU call I: the round's evidence on the final build: whole-benchmark A/B against the round-4 kernels (ab_old), kernel trace,
# PMC traffic (fetch / write passes), per-shape PMC of both Winograd kernels, GPU suite, default bench.
cd "$(dirname "$0")/.."
O=gpurun_out/r5i; mkdir -p $O
rm -f gpurun_out/ab_bench.txt
timeout 900 bash tools/ab_bench.sh run --steps 2 --warmup 1 --no-shard-sim > $O/ab_bench.log 2>&1; cp gpurun_out/ab_bench.txt $O/; cat $O/ab_bench.txt
timeout 1500 bash tools/profile_round.sh r5 > $O/profile_round.log 2>&1; tail -25 $O/profile_round.log
timeout 600 python tools/profile_conv_shapes.py 2>&1 | grep -v amdgpu > $O/insitu_shapes.txt; head -5 $O/insitu_shapes.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; head -c 300 $O/bench_default.json
