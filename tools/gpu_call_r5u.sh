#!/bin/bash
# Round 5, call U: kernel traces of the Python host and of the C-side evaluator (--host ctx) on ONE box: which kernels differ?
R=/root/repo; O=$R/gpurun_out/r5u; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for h in python ctx; do
  B="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-decode --no-max-clip --no-kernel-events --no-shard-sim --no-other-configs --host $h"
  timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_$h -o t -- $B > $O/prof_$h.log 2>&1
  DB=$(find $O/prof_$h -name "*.db" | head -1)
  python $R/tools/rocpd_summary.py $DB > $O/kernel_trace_$h.md 2>&1
  find $O/prof_$h -name "*.db" -delete
done
head -30 $O/kernel_trace_ctx.md
