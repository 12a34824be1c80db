#!/bin/bash
# A/B of two source trees on ONE GPU box (boxes of the pool differ by +-3 %, so numbers from different gpurun calls are not
# comparable).  In the build container:   bash tools/ab_bench.sh prepare [<git rev>]   (exports <rev>, default HEAD, to ab_old/
# and builds its library);  on the GPU box:   bash tools/ab_bench.sh run [bench.py args]   (alternates old / new, 2 rounds).
set -e
cd "$(dirname "$0")/.."
if [ "$1" = "prepare" ]; then
    rev=${2:-HEAD}
    rm -rf ab_old && mkdir ab_old
    git archive "$rev" | tar -x -C ab_old
    (cd ab_old && ./build_lib.sh > /dev/null 2>&1 && echo "ab_old = $(git -C .. rev-parse --short $rev) built")
    exit 0
fi
shift || true
mkdir -p gpurun_out
for round in 1 2; do
    for side in old new; do
        dir=$([ $side = old ] && echo ab_old || echo .)
        v=$(cd $dir && timeout 600 python bench.py --no-cpu-baseline --no-max-clip --no-decode --no-kernel-events $(grep -q -- --no-other-configs bench.py && echo --no-other-configs) "$@" 2>/dev/null | tail -1 |
            python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
        echo "round $round $side: $v" | tee -a gpurun_out/ab_bench.txt
    done
done
