#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2) rocpd sqlite database into a per-kernel table (markdown/CSV-ish):
    python tools/rocpd_summary.py gpurun_out/prof/x_results.db [--skip-first-fraction 0.5]
rocprofv3 --kernel-trace --stats on this image writes sqlite instead of CSV; this reads the `kernels` view."""
import sqlite3
import sys


def main():
    path = sys.argv[1]
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, end, grid_x, workgroup_x, vgpr_count, accum_vgpr_count, lds_size from kernels order by start").fetchall()
    if not rows:
        print("no kernel rows")
        return
    frac = float(sys.argv[sys.argv.index("--skip-first-fraction") + 1]) if "--skip-first-fraction" in sys.argv else 0.0
    t_first, t_last = rows[0][1], rows[-1][2]
    cut = t_first + (t_last - t_first) * frac
    agg = {}
    for name, s, e, gx, wx, vg, ag, lds in rows:
        if s < cut:
            continue
        short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        a = agg.setdefault(short, [0, 0, 10**18, 0, vg, ag, lds])
        d = e - s
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    span = t_last - cut
    print(f"# {path}: {sum(a[0] for a in agg.values())} dispatches, kernel time {tot/1e6:.2f} ms over a {span/1e6:.2f} ms span "
          f"(GPU busy {tot/span*100:.1f}%)")
    print("| kernel | calls | total ms | avg us | min us | max us | % | vgpr | agpr | lds |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| {k[:90]} | {a[0]} | {a[1]/1e6:.3f} | {a[1]/a[0]/1e3:.1f} | {a[2]/1e3:.1f} | {a[3]/1e3:.1f} | {a[1]/tot*100:.1f} | {a[4]} | {a[5]} | {a[6]} |")


if __name__ == "__main__":
    main()
