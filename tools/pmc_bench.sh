#!/bin/bash
# GPU box: HBM-traffic counters for the benchmark command (separate --pmc passes, kernel-trace only) + a
# calibration pass on a pure streaming GEMM whose byte count is known (l0_out: reads 839 MB, writes 210 MB).
OUT=${1:-/root/repo/gpurun_out/pmc_bench}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python /root/repo/bench.py --ddim-steps 2 --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-events --no-overlap --no-decode --no-max-clip --no-shard-sim --no-other-configs"
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o p -- $B > $OUT/fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o p -- $B > $OUT/write.log 2>&1
C="python /root/repo/tools/bench_conv.py --cases l0_out,l0_xo,l0_3x3 --iters 2 --variants 5"
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/cal_fetch -o p -- $C > $OUT/cal_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/cal_write -o p -- $C > $OUT/cal_write.log 2>&1
