#!/bin/bash
# GPU box: per-kernel durations of any command:  bash tools/prof_cmd.sh <tag> <command...>  -> gpurun_out/<tag>_kernels.md
TAG=$1; shift
R=/root/repo; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 600 rocprofv3 --kernel-trace -d $O/${TAG}_prof -o t -- "$@" > $O/${TAG}_prof.log 2>&1)
DB=$(find $O/${TAG}_prof -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $DB > $O/${TAG}_kernels.md 2>&1
find $O/${TAG}_prof -name "*.db" -delete
head -${HEADN:-25} $O/${TAG}_kernels.md | cut -c1-170
