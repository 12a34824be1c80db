#!/bin/bash
# GPU box: fabric-side read traffic (FETCH_SIZE x2 on gfx950; the TCC_HIT/MISS counters abort rocprofv3 on this image) per launch of the Winograd and the direct split conv, per shape.
#   bash tools/pmc_wino_fetch.sh [outdir]
OUT=${1:-/root/repo/gpurun_out/pmc_wino_fetch}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/f -o p -- python /root/repo/tools/bench_wino.py --iters 1 > $OUT/f.log 2>&1
cd /root/repo
python - "$OUT" <<'PY'
import collections, glob, sqlite3, sys
out = sys.argv[1]
db = sqlite3.connect(glob.glob(out + "/f/**/*.db", recursive=True)[0])
cols = [r[1] for r in db.execute("pragma table_info('counters_collection')")]
ci = {c: i for i, c in enumerate(cols)}
name_col = "kernel_name" if "kernel_name" in ci else "name"
per = collections.OrderedDict()
for r in db.execute("select * from counters_collection order by dispatch_id"):
    kn = str(r[ci[name_col]]).replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
    if "conv3x3" not in kn:
        continue
    key = (r[ci["dispatch_id"]], kn, r[ci["grid_size"]])
    per.setdefault(key, {})
    per[key][r[ci["counter_name"]]] = per[key].get(r[ci["counter_name"]], 0.0) + float(r[ci["value"]])
print("# per launch, in launch order of tools/bench_wino.py --iters 1 (per shape: direct x4 [3 warm-up + 1], winograd x4, twice)")
last = None
for (did, kn, grid), c in per.items():
    row = (kn, grid, round(2 * c.get("FETCH_SIZE", 0) * 1024 / 1e6), 0.0)
    if row != last:
        print(f"{kn:28s} grid {grid:8d}  fabric reads {row[2]:7d} MB")
    last = row
PY
find $OUT -name "*.db" -delete
