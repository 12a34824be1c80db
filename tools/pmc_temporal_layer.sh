#!/bin/bash
# GPU box: SQ counters of the fused 64-channel temporal layer, 32 x 32 kernel (WMODE 3) vs window-tiled (WMODE 4), at 200 frames x 4096 pixels
# (tools/bench_temporal_layer.py), two --pmc passes (8 SQ slots each), kernel-trace only.   bash tools/pmc_temporal_layer.sh [outdir]
OUT=${1:-/root/repo/gpurun_out/pmc_temporal}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace \
  -d $OUT/run1 -o p -- python /root/repo/tools/bench_temporal_layer.py > $OUT/run1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM --kernel-trace \
  -d $OUT/run2 -o p -- python /root/repo/tools/bench_temporal_layer.py > $OUT/run2.log 2>&1
cd /root/repo
python - "$OUT" <<'PY'
import collections, glob, sqlite3, sys
out = sys.argv[1]
rows = collections.OrderedDict()
for run in ("run1", "run2"):
    db = sqlite3.connect(glob.glob(f"{out}/{run}/**/*.db", recursive=True)[0])
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    view = "counters_collection" if "counters_collection" in tabs else [t for t in tabs if "counters_collection" in t][0]
    cols = [r[1] for r in db.execute(f"pragma table_info('{view}')")]
    ci = {c: i for i, c in enumerate(cols)}
    nm = "kernel_name" if "kernel_name" in ci else "name"
    per = collections.OrderedDict()
    for r in db.execute(f"select * from {view} order by dispatch_id"):
        kn = str(r[ci[nm]])
        if "temporal_layer" not in kn:
            continue
        d = per.setdefault((kn.split("(")[0][:70], r[ci["dispatch_id"]]), {})
        d[r[ci["counter_name"]]] = d.get(r[ci["counter_name"]], 0.0) + float(r[ci["value"]])
    for (kn, _), d in per.items():
        g = d.get("GRBM_GUI_ACTIVE")
        rows.setdefault(kn, collections.defaultdict(list))
        for c, v in d.items():
            rows[kn][c].append(v)
lines = ["# SQ counters per launch of the fused temporal layer (medians over the launches of tools/bench_temporal_layer.py; sums over the chip)", ""]
for kn, d in rows.items():
    lines.append(f"## {kn}")
    for c, v in sorted(d.items()):
        v = sorted(v)
        lines.append(f"- {c}: median {v[len(v) // 2]:.4g} over {len(v)} launches (min {v[0]:.4g}, max {v[-1]:.4g})")
    lines.append("")
open(out + "/pmc_temporal_layer.md", "w").write("\n".join(lines))
print("\n".join(lines))
PY
