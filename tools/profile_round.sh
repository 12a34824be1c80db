#!/bin/bash
# GPU box: the round's profiling evidence in one call (each rocprofv3 pass is kernel-trace (+ one --pmc set) only):
#   bash tools/profile_round.sh r2     ->  gpurun_out/<tag>_kernel_trace_summary.md, <tag>_pmc_traffic.json,
#                                         <tag>_hbm_by_kernel.md, <tag>_pmc_mfma_util.md, <tag>_pytest.log
TAG=${1:-r2}
HEAD=${2:-unknown}      # the commit of the snapshot under test (the GPU box has no .git): stamped into <tag>_pmc_traffic.json
R=/root/repo; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-decode --no-max-clip --no-kernel-events --no-shard-sim --no-other-configs"
timeout 400 rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof -o t -- $B > $O/${TAG}_prof.log 2>&1
DB=$(find $O/${TAG}_prof -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $DB > $O/${TAG}_kernel_trace_summary.md 2>&1
find $O/${TAG}_prof -name "*.db" -delete
bash $R/tools/pmc_bench.sh $O/${TAG}_pmc > /dev/null 2>&1
F=$(find $O/${TAG}_pmc/fetch -name "*.db" | head -1); W=$(find $O/${TAG}_pmc/write -name "*.db" | head -1)
python $R/tools/pmc_traffic_json.py $F $W $O/${TAG}_pmc_traffic.json $HEAD > $O/${TAG}_pmc_traffic.log 2>&1
python $R/tools/pmc_hbm_by_kernel.py $F $W 2 > $O/${TAG}_hbm_by_kernel.md 2>&1
find $O/${TAG}_pmc -name "*.db" -delete
bash $R/tools/pmc_wino_shapes.sh $O/${TAG}_pmc_wino_shapes > /dev/null 2>&1
cp $O/${TAG}_pmc_wino_shapes/pmc_wino_shapes.md $O/${TAG}_pmc_wino_by_shape.md 2>/dev/null
cd $R
(timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3) > $O/${TAG}_pytest.log
head -12 $O/${TAG}_kernel_trace_summary.md; cat $O/${TAG}_pytest.log; head -5 $O/${TAG}_hbm_by_kernel.md
