R=/root/repo; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-decode --no-max-clip --no-kernel-events"
timeout 400 rocprofv3 --kernel-trace -d $O/tl_prof -o t -- $B > $O/tl_prof.log 2>&1
DB=$(find $O/tl_prof -name "*.db" | head -1)
python $R/tools/rocpd_timeline.py $DB --skip 0.6 > $O/r2_timeline.txt 2>&1
find $O/tl_prof -name "*.db" -delete
cd $R
timeout 300 python tools/profile_conv_shapes.py > $O/r2_conv_shapes.txt 2>&1
cat $O/r2_timeline.txt; head -30 $O/r2_conv_shapes.txt
