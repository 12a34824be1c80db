#!/bin/bash
# GPU box: matrix-pipe utilisation of the split-operand conv kernels from PMC counters (one --pmc set, kernel-trace only).
#   bash tools/pmc_mfma_util.sh [outdir]   -> <outdir>/mfma_util.md
# SQ_VALU_MFMA_BUSY_CYCLES counts cycles in which a SIMD's matrix pipe is busy (32 per v_mfma_f32_32x32x16_bf16, 16 per
# v_mfma_f32_16x16x32_bf16: the same per flop),
# summed over all 1024 SIMDs; GRBM_GUI_ACTIVE = clock cycles of the kernel summed over the 8 XCDs (checked: 4.69e6 for a
# 290 us launch = 8 x 586k cycles at 2.02 GHz).  utilisation = BUSY / (GUI_ACTIVE / 8 * 1024 SIMDs).
OUT=${1:-/root/repo/gpurun_out/pmc_mfma}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace \
  -d $OUT/run -o p -- python /root/repo/tools/bench_conv.py --cases l0_3x3,l0_3x3_cat,l1_3x3,l2_3x3,l3_3x3 --iters 3 --variants 0x100580D,0x580D > $OUT/run.log 2>&1
cd /root/repo
python - "$OUT" <<'PY'
import collections, glob, sqlite3, sys
out = sys.argv[1]
db = sqlite3.connect(glob.glob(out + "/run/**/*.db", recursive=True)[0])
cols = [r[1] for r in db.execute("pragma table_info('counters_collection')")]
ci = {c: i for i, c in enumerate(cols)}
name_col = "kernel_name" if "kernel_name" in ci else "name"
did = "dispatch_id" if "dispatch_id" in ci else None
per = collections.defaultdict(dict)
for r in db.execute("select * from counters_collection"):
    kn = str(r[ci[name_col]]).replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if "conv3x3_bf16" not in kn and "conv3x3_wino" not in kn:
        continue
    key = (kn, r[ci[did]] if did else 0, r[ci["grid_size"]] if "grid_size" in ci else 0)
    per[key][r[ci["counter_name"]]] = per[key].get(r[ci["counter_name"]], 0.0) + float(r[ci["value"]])
lines = ["# Matrix-pipe utilisation of the split-operand 3x3 conv kernel (PMC, `tools/pmc_mfma_util.sh`; one row per dispatch of "
         "`tools/bench_conv.py --variants 0x100580D,0x580D`: the shipped kernel on v_mfma_f32_16x16x32_bf16 = template argument `true`, then the 32x32x16 form)", "",
         "| kernel | grid | MFMA busy cycles (sum over SIMDs) | MFMA insts | VALU insts | GUI_ACTIVE cycles (sum over 8 XCDs) | shader clock cycles | matrix pipe busy = MFMA busy / (cycles x 1024 SIMDs) |",
         "|---|---|---|---|---|---|---|---|"]
for (kn, d, g), c in sorted(per.items(), key=lambda kv: kv[0][1]):
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    cyc = gui / 8.0
    util = busy / (cyc * 1024) if gui else float("nan")
    lines.append(f"| {kn} | {g} | {busy:.4g} | {c.get('SQ_INSTS_MFMA', 0):.4g} | {c.get('SQ_INSTS_VALU', 0):.4g} | {gui:.4g} | {cyc:.4g} | {util:.3f} |")
open(out + "/mfma_util.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
find $OUT -name "*.db" -delete
