#!/bin/bash
# GPU box: matrix-pipe utilisation and instruction counts of conv3x3_wino_kernel (the shipped dominant kernel) at EVERY 3x3 launch
# shape of one denoiser evaluation (tools/bench_wino.py SHAPES = SURVEY A.5), from PMC counters (one --pmc set, kernel-trace only).
#   bash tools/pmc_wino_shapes.sh [outdir]   -> <outdir>/pmc_wino_shapes.md
# SQ_VALU_MFMA_BUSY_CYCLES = cycles a SIMD's matrix pipe is busy (16 per v_mfma_f32_16x16x32_bf16), summed over the 1024 SIMDs;
# GRBM_GUI_ACTIVE = clock cycles of the launch summed over the 8 XCDs.  busy = MFMA_BUSY / (GUI_ACTIVE / 8 x 1024).
# executed flops = SQ_INSTS_MFMA x 16384; frac_executed at the nominal peak = executed / time / 2.5 PF (time = GUI_ACTIVE / 8 / clock is not
# used: the HIP-event time of the same launches is printed by bench_wino.py itself into run.log).
OUT=${1:-/root/repo/gpurun_out/pmc_wino_shapes}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace \
  -d $OUT/run -o p -- python /root/repo/tools/bench_wino.py --wino-only --iters 1 > $OUT/run.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace \
  -d $OUT/run4 -o p -- python /root/repo/tools/bench_wino.py --wino4 --iters 1 --only 0 1 2 3 > $OUT/run4.log 2>&1
cd /root/repo
python - "$OUT" <<'PY'
import collections, glob, sqlite3, sys
sys.path.insert(0, "tools")
out = sys.argv[1]
SHAPES = [(200, 64, 64, 64, 0, 64), (200, 64, 64, 64, 64, 64), (200, 32, 32, 64, 0, 128), (200, 32, 32, 128, 0, 128), (200, 32, 32, 128, 128, 64),
          (200, 32, 32, 64, 0, 64), (200, 16, 16, 128, 0, 256), (200, 16, 16, 256, 0, 256), (200, 16, 16, 256, 256, 128), (200, 16, 16, 128, 0, 128),
          (200, 8, 8, 256, 0, 512), (200, 8, 8, 512, 0, 512), (200, 8, 8, 512, 512, 256), (200, 8, 8, 256, 0, 256)]
db = sqlite3.connect(glob.glob(out + "/run/**/*.db", recursive=True)[0])
cols = [r[1] for r in db.execute("pragma table_info('counters_collection')")]
ci = {c: i for i, c in enumerate(cols)}
name_col = "kernel_name" if "kernel_name" in ci else "name"
per = collections.OrderedDict()
for r in db.execute("select * from counters_collection order by dispatch_id"):
    kn = str(r[ci[name_col]])
    if "conv3x3_wino" not in kn:
        continue
    per.setdefault(r[ci["dispatch_id"]], {})
    per[r[ci["dispatch_id"]]][r[ci["counter_name"]]] = per[r[ci["dispatch_id"]]].get(r[ci["counter_name"]], 0.0) + float(r[ci["value"]])
disp = list(per.values())
n_per = len(disp) // len(SHAPES)
lines = ["# conv3x3_wino_kernel per launch shape (PMC, `tools/pmc_wino_shapes.sh`: one rocprofv3 --pmc pass over `tools/bench_wino.py --wino-only --iters 1`; "
         f"{len(disp)} dispatches = {n_per} per shape, medians)", "",
         "| M | N | K | tiles | MFMA insts | VALU insts | VALU / MFMA | MFMA busy cycles (sum over SIMDs) | shader cycles (GUI_ACTIVE / 8) | matrix pipe busy | executed TFLOP at 16384 flop per MFMA | algorithmic GFLOP (2 M N K) | executed / algorithmic |",
         "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
def med(v):
    v = sorted(v)
    return v[len(v) // 2]
for si, (F, H, W, C0, C1, N) in enumerate(SHAPES):
    g = disp[si * n_per:(si + 1) * n_per]
    if not g:
        continue
    m = {k: med([d.get(k, 0.0) for d in g]) for k in g[0]}
    M, K = F * H * W, 9 * (C0 + C1)
    cyc = m.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    busy = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    ex = m.get("SQ_INSTS_MFMA", 0.0) * 16384.0
    alg = 2.0 * M * N * K
    lines.append(f"| {M} | {N} | {K} | {M // 256 * (N // 64)} | {m.get('SQ_INSTS_MFMA', 0):.4g} | {m.get('SQ_INSTS_VALU', 0):.4g} | "
                 f"{m.get('SQ_INSTS_VALU', 0) / max(m.get('SQ_INSTS_MFMA', 1), 1):.2f} | {busy:.4g} | {cyc:.4g} | {busy / (cyc * 1024) if cyc else float('nan'):.3f} | "
                 f"{ex / 1e12:.3f} | {alg / 1e9:.2f} | {ex / alg:.3f} |")
# ---- the F(4x4,3x3) kernel on the four level-0 / level-1 shapes its geometry takes (shipped for the first one only)
db4s = glob.glob(out + "/run4/**/*.db", recursive=True)
if db4s:
    db4 = sqlite3.connect(db4s[0])
    per4 = collections.OrderedDict()
    for r in db4.execute("select * from counters_collection order by dispatch_id"):
        if "conv3x3_wino4" not in str(r[ci[name_col]]):
            continue
        per4.setdefault(r[ci["dispatch_id"]], {})
        per4[r[ci["dispatch_id"]]][r[ci["counter_name"]]] = per4[r[ci["dispatch_id"]]].get(r[ci["counter_name"]], 0.0) + float(r[ci["value"]])
    d4 = list(per4.values())
    n4 = max(1, len(d4) // 4)
    lines += ["", f"## conv3x3_wino4_kernel (F(4x4,3x3); {len(d4)} dispatches = {n4} per shape, medians; executed / algorithmic = 8 cross terms x 36/144 = 2.0 by construction)", "",
              "| M | N | K | tiles | MFMA insts | VALU insts | VALU / MFMA | MFMA busy cycles | shader cycles | matrix pipe busy | executed TFLOP | algorithmic GFLOP | executed / algorithmic |",
              "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for si, (F, H, W, C0, C1, N) in enumerate(SHAPES[:4]):
        g = d4[si * n4:(si + 1) * n4]
        if not g:
            continue
        m = {k: med([d.get(k, 0.0) for d in g]) for k in g[0]}
        M, K = F * H * W, 9 * (C0 + C1)
        cyc = m.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        busy = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        ex = m.get("SQ_INSTS_MFMA", 0.0) * 16384.0
        alg = 2.0 * M * N * K
        lines.append(f"| {M} | {N} | {K} | {M // 256 * (N // 64)} | {m.get('SQ_INSTS_MFMA', 0):.4g} | {m.get('SQ_INSTS_VALU', 0):.4g} | "
                     f"{m.get('SQ_INSTS_VALU', 0) / max(m.get('SQ_INSTS_MFMA', 1), 1):.2f} | {busy:.4g} | {cyc:.4g} | {busy / (cyc * 1024) if cyc else float('nan'):.3f} | "
                     f"{ex / 1e12:.3f} | {alg / 1e9:.2f} | {ex / alg:.3f} |")
lines += ["", "(executed / algorithmic = 6 cross terms x 16/36 Winograd multiplies = 2.667 by construction: a check that the counters see the whole launch;",
          " the HIP-event time of the same launches is in run.log -- frac_executed = executed TFLOP / time / 2.5 PFLOP/s)"]
open(out + "/pmc_wino_shapes.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
cp $OUT/run.log $OUT/bench_wino_under_pmc.log 2>/dev/null
cat $OUT/run4.log >> $OUT/bench_wino_under_pmc.log 2>/dev/null
find $OUT -name "*.db" -delete
