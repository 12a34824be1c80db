R=/root/repo; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-decode --no-max-clip --no-kernel-events --no-shard-sim --no-other-configs"
timeout 400 rocprofv3 --kernel-trace --stats -d $O/r3b_prof -o t -- $B > $O/r3b_prof.log 2>&1
DB=$(find $O/r3b_prof -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $DB > $O/r3b_kernel_trace_summary.md 2>&1
find $O/r3b_prof -name "*.db" -delete
head -30 $O/r3b_kernel_trace_summary.md | cut -c1-150
