#!/usr/bin/env python3
"""Per-class comparison of two kernel-trace summaries (tools/rocpd_summary.py output): time per denoiser evaluation and share, by kernel
class.   python tools/class_table.py <summary A.md> <evaluations A> <frames A> <summary B.md> <evaluations B> <frames B>"""
import re, sys

CLASSES = [("3x3 ResBlock convs (Winograd F(2x2), F(4x4), direct)", ("conv3x3_",)),
           ("fused temporal layer, 64 channels", ("temporal_layer13_kernel", "temporal_layer16_kernel", "temporal_layer_c64_bf16_kernel<4, 0, true, true", "temporal_layer_c64_kernel")),
           ("temporal attention core, C >= 128 (EXT form, fp32 form)", ("temporal_layer_c64_bf16_kernel", "temporal_attn_kernel")),
           ("1x1 projections / resampling convs (gemm1x1_*, conv_gemm_*)", ("gemm1x1_", "conv_gemm_")),
           ("spatial linear attention (64-channel fused + core)", ("sla_",)),
           ("cross-attention", ("xattn_",)),
           ("GroupNorm apply / LayerNorm statistics", ("gn_", "ln_rowstats")),
           ("init conv, heads, mid spatial attention, linear", ("init_conv", "head_out", "frame_attn", "linear_kernel", "sinusoidal")),
           ("sampler (x0, quantile select, update, noise)", ("ddim_", "select_", "philox", "cfg_combine")),
           ("torch / rocBLAS / copy kernels of weight packing and clip preparation (whole process, outside the timed loop; per evaluation here)", ("at::native", "__amd_rocclr", "Cijk_"))]


def load(path):
    rows = []
    for line in open(path):
        m = re.match(r"\| (.+?) \| (\d+) \| ([0-9.]+) \| ([0-9.]+) \|", line)
        if m and m.group(1) != "kernel":
            rows.append((m.group(1), int(m.group(2)), float(m.group(3))))
    return rows


def table(rows, evals):
    out, rest = [], list(rows)
    for name, keys in CLASSES:
        hit = [r for r in rest if any(k in r[0] for k in keys)]
        rest = [r for r in rest if r not in hit]
        out.append((name, sum(r[2] for r in hit) / evals, sum(r[1] for r in hit) / evals))
    out.append(("other", sum(r[2] for r in rest) / evals, sum(r[1] for r in rest) / evals))
    return out


a, ea, fa, b, eb, fb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5]), int(sys.argv[6])
ta, tb = table(load(a), ea), table(load(b), eb)
sa, sb = sum(t[1] for t in ta), sum(t[1] for t in tb)
print(f"| class | A: ms / evaluation | A: share | A: us / frame | A: launches / evaluation | B: ms / evaluation | B: share | B: us / frame | B: launches / evaluation | (A us/frame x 4) / (B us/frame) |")
print("|---|---|---|---|---|---|---|---|---|---|")
for (n, ma, la), (_, mb, lb) in zip(ta, tb):
    ua, ub = ma * 1e3 / fa, mb * 1e3 / fb
    print(f"| {n} | {ma:.3f} | {100 * ma / sa:.1f} % | {ua:.2f} | {la:.0f} | {mb:.3f} | {100 * mb / sb:.1f} % | {ub:.2f} | {lb:.0f} | {4 * ua / ub if ub else 0:.2f} |")
print(f"| **total** | {sa:.3f} | | {sa * 1e3 / fa:.2f} | | {sb:.3f} | | {sb * 1e3 / fb:.2f} | | {4 * (sa / fa) / (sb / fb):.2f} |")
