#!/bin/bash
# Round 5, GPU call G: the whole GPU suite + smoke + default bench on the build that ships the F(4x4) form for the 64-channel level-0 convs.
cd "$(dirname "$0")/.."
O=gpurun_out/r5g; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee $O/rc.txt
tail -4 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/rc.txt; tail -3 $O/smoke.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" | tee -a $O/rc.txt
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5g/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["kernel"][:40], d["roofline"]["frac"], d["roofline"]["avg_launch_us"])
for r in d.get("roofline_other", []): print("  ", r["kernel"][:40], r["frac"], r["avg_launch_us"], r["share_of_conv_time"])
print(d["shard_sim"]["shard_overhead"], [o.get("frames_per_s") for o in d["other_configs"]], d["cpu_baseline"]["value"])
PY
wc -l $O/bench_default.json
