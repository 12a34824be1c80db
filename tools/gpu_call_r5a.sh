#!/bin/bash
# Round 5, GPU call A: correctness of the round's first kernel changes + A/B against the round-4 tree + evidence runs.
cd "$(dirname "$0")/.."
O=gpurun_out/r5a; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee $O/rc.txt
timeout 120 tools/ubench/wino_stream.bin > $O/wino_stream.txt 2>&1
timeout 300 python tools/bench_wino.py --iters 10 > $O/bench_wino_new.txt 2>&1
(cd ab_old && timeout 300 python tools/bench_wino.py --iters 10) > $O/bench_wino_old.txt 2>&1
rm -f gpurun_out/ab_bench.txt gpurun_out/ab_policy.txt
timeout 900 bash tools/ab_bench.sh run --steps 2 --warmup 1 --no-shard-sim > $O/ab_bench.log 2>&1
timeout 600 bash tools/ab_policy.sh "0x300580D 0x700580D" 2 > $O/ab_policy.log 2>&1
cp gpurun_out/ab_bench.txt gpurun_out/ab_policy.txt $O/ 2>/dev/null
timeout 600 bash tools/pmc_wino_shapes.sh /root/repo/$O/pmc_wino_shapes > $O/pmc_wino_shapes.log 2>&1
timeout 600 python bench.py --steps 3 --warmup 1 > $O/bench_default.json 2> $O/bench_default.err
tail -3 $O/pytest_gpu.log; cat $O/ab_bench.txt $O/ab_policy.txt; head -c 600 $O/bench_default.json
