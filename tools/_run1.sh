cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hip_ops.py -q -x -k "winograd or wino or families" 2>&1 | tail -2
for r in 1 2; do
(cd ab_old && timeout 200 python tools/bench_wino.py --iters 20 --wino-only) > gpurun_out/r4w_iso_old$r.txt 2>&1
timeout 200 python tools/bench_wino.py --iters 20 --wino-only > gpurun_out/r4w_iso_new$r.txt 2>&1
done
paste <(grep winograd gpurun_out/r4w_iso_old1.txt | awk '{print $1,$2,$3,$4, $6}') <(grep winograd gpurun_out/r4w_iso_new1.txt | awk '{print $6}') <(grep winograd gpurun_out/r4w_iso_old2.txt | awk '{print $6}') <(grep winograd gpurun_out/r4w_iso_new2.txt | awk '{print $6}') | tee gpurun_out/r4w_iso_table.txt
rm -f gpurun_out/ab_bench.txt
bash tools/ab_bench.sh run --no-shard-sim
