#!/bin/bash
# Round 5, GPU call S: the round's evidence on the FINAL build: kernel trace, PMC traffic (fetch / write passes), per-shape PMC of both Winograd kernels,
# GPU suite, smoke, in-situ conv shapes, default bench.
cd "$(dirname "$0")/.."
O=gpurun_out/r5s; mkdir -p $O
timeout 1500 bash tools/profile_round.sh r5 > $O/profile_round.log 2>&1; tail -25 $O/profile_round.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 600 python tools/profile_conv_shapes.py 2>&1 | grep -v amdgpu > $O/insitu_shapes.txt; head -5 $O/insitu_shapes.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; head -c 300 $O/bench_default.json
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -3 > $O/pytest_full.log; cat $O/pytest_full.log
