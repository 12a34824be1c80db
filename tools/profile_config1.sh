#!/bin/bash
# GPU box: BASELINE configs[1] (128 x 128, 400 frames) under the profiler, with configs[2] (256 x 256, 200 frames) on the same box for the
# per-class comparison: kernel traces, the in-situ GEMM shape table, FETCH_SIZE / WRITE_SIZE passes (each rocprofv3 pass: kernel-trace (+ one
# --pmc counter) only).   bash tools/profile_config1.sh <tag>   -> gpurun_out/<tag>_config1_*
TAG=${1:-r6}
R=/root/repo; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
COMMON="--steps 1 --warmup 0 --no-cpu-baseline --no-decode --no-max-clip --no-kernel-events --no-shard-sim --no-other-configs"
timeout 400 rocprofv3 --kernel-trace --stats -d $O/${TAG}_c1_prof -o t -- python $R/bench.py --res 128 --frames 400 $COMMON > $O/${TAG}_c1_prof.log 2>&1
python $R/tools/rocpd_summary.py $(find $O/${TAG}_c1_prof -name "*.db" | head -1) > $O/${TAG}_config1_kernel_trace_summary.md 2>&1
find $O/${TAG}_c1_prof -name "*.db" -delete
timeout 400 rocprofv3 --kernel-trace --stats -d $O/${TAG}_c2_prof -o t -- python $R/bench.py $COMMON > $O/${TAG}_c2_prof.log 2>&1
python $R/tools/rocpd_summary.py $(find $O/${TAG}_c2_prof -name "*.db" | head -1) > $O/${TAG}_kernel_trace_summary.md 2>&1
find $O/${TAG}_c2_prof -name "*.db" -delete
P="--ddim-steps 2 --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-events --no-overlap --no-decode --no-max-clip --no-shard-sim --no-other-configs"
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/${TAG}_c1_pmc/fetch -o p -- python $R/bench.py --res 128 --frames 400 $P > $O/${TAG}_c1_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/${TAG}_c1_pmc/write -o p -- python $R/bench.py --res 128 --frames 400 $P > $O/${TAG}_c1_write.log 2>&1
F=$(find $O/${TAG}_c1_pmc/fetch -name "*.db" | head -1); W=$(find $O/${TAG}_c1_pmc/write -name "*.db" | head -1)
python $R/tools/pmc_traffic_json.py $F $W $O/${TAG}_config1_pmc_traffic.json > $O/${TAG}_config1_pmc_traffic.log 2>&1
python $R/tools/pmc_hbm_by_kernel.py $F $W 2 > $O/${TAG}_config1_hbm_by_kernel.md 2>&1
find $O/${TAG}_c1_pmc -name "*.db" -delete
cd $R
timeout 300 python tools/profile_conv_shapes.py --frames 400 --res 128 > $O/${TAG}_config1_insitu_shapes.txt 2>&1
timeout 300 python tools/profile_conv_shapes.py > $O/${TAG}_insitu_shapes.txt 2>&1
python tools/class_table.py $O/${TAG}_config1_kernel_trace_summary.md 50 400 $O/${TAG}_kernel_trace_summary.md 50 200 > $O/${TAG}_config1_vs_config2_by_class.md 2>&1
cat $O/${TAG}_config1_vs_config2_by_class.md; head -14 $O/${TAG}_config1_kernel_trace_summary.md; head -12 $O/${TAG}_config1_insitu_shapes.txt; tail -3 $O/${TAG}_config1_pmc_traffic.log
