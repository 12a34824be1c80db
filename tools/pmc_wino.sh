#!/bin/bash
# GPU box: SQ stall / LDS / matrix-pipe counters of the Winograd split conv and the direct one (one --pmc set per pass, kernel-trace only).
#   bash tools/pmc_wino.sh [outdir] [shape indices of tools/bench_wino.py]   -> <outdir>/pmc_wino.md
OUT=${1:-/root/repo/gpurun_out/pmc_wino}; shift; SH=${@:-0 1 3}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
timeout 300 rocprofv3 --pmc $P1 --kernel-trace -d $OUT/p1 -o p -- python /root/repo/tools/bench_wino.py --iters 2 --only $SH > $OUT/p1.log 2>&1
timeout 300 rocprofv3 --pmc $P2 --kernel-trace -d $OUT/p2 -o p -- python /root/repo/tools/bench_wino.py --iters 2 --only $SH > $OUT/p2.log 2>&1
cd /root/repo
python - "$OUT" <<'PY'
import collections, glob, sqlite3, sys
out = sys.argv[1]
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for p in ("p1", "p2"):
    dbs = glob.glob(out + f"/{p}/**/*.db", recursive=True)
    if not dbs:
        continue
    db = sqlite3.connect(dbs[0])
    cols = [r[1] for r in db.execute("pragma table_info('counters_collection')")]
    ci = {c: i for i, c in enumerate(cols)}
    name_col = "kernel_name" if "kernel_name" in ci else "name"
    per = collections.defaultdict(dict)
    for r in db.execute("select * from counters_collection"):
        kn = str(r[ci[name_col]]).replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if "conv3x3" not in kn:
            continue
        key = (kn, r[ci["dispatch_id"]], r[ci["grid_size"]])
        per[key][r[ci["counter_name"]]] = per[key].get(r[ci["counter_name"]], 0.0) + float(r[ci["value"]])
    for (kn, did, grid), c in per.items():
        for k, v in c.items():
            rows[(kn.split("<")[0], grid)][k].append(v)
lines = ["# SQ counters per launch (median over the launches of tools/bench_wino.py), `tools/pmc_wino.sh`", ""]
for (kn, grid), c in sorted(rows.items()):
    med = {k: sorted(v)[len(v) // 2] for k, v in c.items()}
    lines.append(f"## {kn}  grid {grid}  ({len(next(iter(c.values())))} launches)")
    for k in sorted(med):
        lines.append(f"  {k:28s} {med[k]:16.0f}")
    wc = med.get("SQ_WAVE_CYCLES")
    if wc:
        lines.append("  -- fractions of SQ_WAVE_CYCLES: wait_any %.3f  wait_inst_any %.3f  (of which LDS %.3f)  active %.3f" % (
            med.get("SQ_WAIT_ANY", 0) / wc, med.get("SQ_WAIT_INST_ANY", 0) / wc, med.get("SQ_WAIT_INST_LDS", 0) / wc, med.get("SQ_ACTIVE_INST_ANY", 0) / wc))
    if med.get("SQ_LDS_IDX_ACTIVE"):
        lines.append("  -- LDS bank-conflict cycles / LDS active cycles: %.3f" % (med.get("SQ_LDS_BANK_CONFLICT", 0) / med["SQ_LDS_IDX_ACTIVE"]))
    lines.append("")
open(out + "/pmc_wino.md", "w").write("\n".join(lines))
print("\n".join(lines))
PY
find $OUT -name "*.db" -delete
