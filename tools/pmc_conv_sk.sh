#!/bin/bash
# GPU box: cycles, matrix-pipe busy and wave-state counters of the 3x3 split conv kernels (v2 = policy 0x580D, stream-K = 0x5C0D),
# one --pmc set per pass, kernel-trace only.   bash tools/pmc_conv_sk.sh [outdir] [cases] [variants]
# GRBM_GUI_ACTIVE / 8 = shader-clock cycles of a launch (the chip clocks to its power budget: wall time alone hides whether a
# change saved cycles or cost clock); SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs) = matrix-pipe busy; SQ_WAIT_* in quad-cycles.
OUT=${1:-/root/repo/gpurun_out/pmc_conv_sk}; CASES=${2:-l0_3x3,l0_3x3_cat,l1_3x3,l3_3x3}; VARS=${3:-0x580D,0x5C0D}
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
C="python /root/repo/tools/bench_conv.py --gn --cases $CASES --iters 3 --variants $VARS"
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 250 rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -o p -- $C > $OUT/p$i.log 2>&1
done
cd /root/repo
python - "$OUT" <<'PY'
import collections, glob, sqlite3, sys
out = sys.argv[1]
per = collections.defaultdict(dict)
for dbf in sorted(glob.glob(out + "/p*/**/*.db", recursive=True)):
    db = sqlite3.connect(dbf)
    cols = [r[1] for r in db.execute("pragma table_info('counters_collection')")]
    ci = {c: i for i, c in enumerate(cols)}
    name_col = "kernel_name" if "kernel_name" in ci else "name"
    disp = collections.defaultdict(dict)
    for r in db.execute("select * from counters_collection"):
        kn = str(r[ci[name_col]]).replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if "conv3x3" not in kn:
            continue
        d = disp[(r[ci["dispatch_id"]], kn, r[ci["grid_size"]])]
        d[r[ci["counter_name"]]] = d.get(r[ci["counter_name"]], 0.0) + float(r[ci["value"]])
    # dispatches in launch order: bench_conv runs 1 warm-up + 3 timed launches per (variant, case)
    for n, (key, c) in enumerate(sorted(disp.items())):
        per[(n // 4, key[1], key[2])].setdefault("n", 0)
        for k, v in c.items():
            per[(n // 4, key[1], key[2])][k] = per[(n // 4, key[1], key[2])].get(k, 0.0) + v / 4.0 / (2.0 if k == "GRBM_GUI_ACTIVE" else 1.0)
lines = ["| launch group | kernel | grid (threads) | cycles | MFMA busy | VALU / MFMA insts | wave-cycles/SIMD-cycle | WAIT_ANY | WAIT_INST_ANY | ACTIVE_ANY | WAIT_INST_LDS | ACTIVE_VALU | ACTIVE_LDS |", "|" + "---|" * 13]
for (g, kn, grid), c in sorted(per.items()):
    cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    wc = c.get("SQ_WAVE_CYCLES", 0.0) * 4.0
    f = lambda k: (c.get(k, 0.0) * 4.0 / wc) if wc else float("nan")
    lines.append(f"| {g} | {kn} | {grid} | {cyc:.4g} | {c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (cyc * 1024) if cyc else 0:.3f} | "
                 f"{c.get('SQ_INSTS_VALU', 0) / max(1.0, c.get('SQ_INSTS_MFMA', 0)):.2f} | {wc / (cyc * 1024) if cyc else 0:.2f} | {f('SQ_WAIT_ANY'):.3f} | "
                 f"{f('SQ_WAIT_INST_ANY'):.3f} | {f('SQ_ACTIVE_INST_ANY'):.3f} | {f('SQ_WAIT_INST_LDS'):.3f} | {f('SQ_ACTIVE_INST_VALU'):.3f} | {f('SQ_ACTIVE_INST_LDS'):.3f} |")
open(out + "/conv_sk_pmc.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
grep -h "us " $OUT/p1.log | grep -v amdgpu | head -20
find $OUT -name "*.db" -delete
