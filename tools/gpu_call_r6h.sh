#!/bin/bash
# Round 6, call H: the default bench line (what the driver runs) + the Python-host / C-host / graph A/B at the benchmark clip.
cd "$(dirname "$0")/.."
O=gpurun_out/r6h; mkdir -p $O
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6h/bench_default.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("metric", "value", "unit", "ms_per_step", "n_gpus", "dtype")})
r = d["roofline"]
print({k: r.get(k) for k in ("kernel", "avg_launch_us", "frac", "frac_algorithmic", "traffic", "traffic_profile", "traffic_profile_head", "traffic_refused", "algorithmic_bytes_per_launch_avg")})
print("cpu_baseline", d.get("cpu_baseline")); print("other_configs", d.get("other_configs")); print("max_clip_frames", d.get("max_clip_frames"))
print({k: d.get(k) for k in ("shard_sim",)})
PY
