cd /root/repo
for round in 1 2; do
  for f in "" "--no-fuse-h1" "--no-fuse-h1 --no-overlap"; do
    v=$(timeout 300 python bench.py --no-cpu-baseline --no-max-clip --no-decode --no-kernel-events --no-shard-sim --no-other-configs --steps 2 --warmup 1 $f 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(round(d['value'], 2), round(d['ms_per_step'], 1))")
    echo "round $round [$f]: $v" | tee -a gpurun_out/ab_h1.txt
  done
done
