#!/usr/bin/env python3
"""Build profiles/*_pmc_traffic.json from the two PMC passes of tools/pmc_bench.sh:
    python tools/pmc_traffic_json.py <fetch_results.db> <write_results.db> [out.json] [commit the passes ran on]
Sums FETCH_SIZE / WRITE_SIZE (KiB) over every dawn_conv_gemm kernel (conv_gemm_kernel, conv_gemm_glds_kernel,
conv3x3_halo_kernel), applies the gfx950 x2 FETCH_SIZE correction (calibrated, see `calibration`), and reports
HBM bytes per conv launch -- the `traffic` figure bench.py attaches to the roofline object."""
import collections, json, sqlite3, sys


def sums(path, counter):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info('counters_collection')")]
    ci = {c: i for i, c in enumerate(cols)}
    name_col = "kernel_name" if "kernel_name" in ci else "name"
    per = collections.defaultdict(lambda: [0, 0.0])
    for r in db.execute("select * from counters_collection"):
        if r[ci["counter_name"]] != counter:
            continue
        kn = str(r[ci[name_col]]).replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        per[kn][0] += 1
        per[kn][1] += float(r[ci["value"]])
    return per


def kernel_kind(k):
    """kernel name -> the kernel class bench.py's roofline uses (bench.py kind_of)"""
    if k.startswith("conv3x3_wino4"):
        return "conv3x3_wino4"
    if k.startswith("conv3x3_wino"):
        return "conv3x3_wino"
    if k.startswith("conv3x3_bf16") or k.startswith("conv3x3_halo_bf16"):
        return "conv3x3"
    if k.startswith("gemm1x1_rowreg"):
        return "gemm1x1_rowreg"
    if k.startswith("gemm1x1_rowacc"):
        return "gemm1x1_rowacc"
    if k.startswith("gemm1x1_"):
        return "gemm1x1"
    return "fp32"


def main():
    fetch, write = sums(sys.argv[1], "FETCH_SIZE"), sums(sys.argv[2], "WRITE_SIZE")
    is_conv = lambda k: k.startswith("conv_gemm") or k.startswith("conv3x3") or k.startswith("gemm1x1")
    n = sum(v[0] for k, v in fetch.items() if is_conv(k))
    f = sum(v[1] for k, v in fetch.items() if is_conv(k))
    w = sum(v[1] for k, v in write.items() if is_conv(k))
    def per_class(pred):
        nn = sum(v[0] for k, v in fetch.items() if pred(k))
        if not nn:
            return None
        return (2.0 * sum(v[1] for k, v in fetch.items() if pred(k)) + sum(v[1] for k, v in write.items() if pred(k))) / nn * 1024.0

    out = {
        "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace -- python bench.py --ddim-steps 2 --steps 1 "
                   "--warmup 0 --no-cpu-baseline --no-kernel-events --no-overlap (tools/pmc_bench.sh, separate passes)",
        "workload": "256x256, 200 frames",
        "conv_gemm_launches": n,
        "fetch_size_kib_per_launch_raw": f / n,
        "write_size_kib_per_launch": w / n,
        "fetch_correction": 2.0,
        "calibration": "streaming 1x1 GEMM reading 819,200 KiB reports FETCH_SIZE 414-423k KiB (x0.51): the gfx950 "
                       "half-count of MI355X_MICROARCH.md holds for 16 B/lane loads; WRITE_SIZE is exact",
        "hbm_bytes_per_launch": (2.0 * f + w) / n * 1024.0,
        "hbm_bytes_per_launch_split_bf16": per_class(lambda k: is_conv(k) and ("bf16" in k or "wino" in k)),
        "hbm_bytes_per_launch_conv3x3_wino": per_class(lambda k: k.startswith("conv3x3_wino_kernel")),
        "hbm_bytes_per_launch_conv3x3_wino4": per_class(lambda k: k.startswith("conv3x3_wino4")),
        "hbm_bytes_per_launch_conv3x3_bf16": per_class(lambda k: k.startswith("conv3x3_bf16") or k.startswith("conv3x3_halo_bf16")),
        "hbm_bytes_per_launch_gemm1x1_bf16": per_class(lambda k: k.startswith("gemm1x1_bf16")),
        "hbm_bytes_per_launch_gemm1x1_rowreg": per_class(lambda k: k.startswith("gemm1x1_rowreg")),
        "hbm_bytes_per_launch_gemm1x1_rowacc": per_class(lambda k: k.startswith("gemm1x1_rowacc")),
        "hbm_bytes_per_launch_gemm1x1_family": per_class(lambda k: k.startswith("gemm1x1_")),
        "hbm_bytes_per_launch_fp32": per_class(lambda k: is_conv(k) and "bf16" not in k and "wino" not in k and not k.startswith("gemm1x1_")),
        "per_kernel": {k: {"launches": fetch[k][0], "fetch_kib_raw_avg": fetch[k][1] / fetch[k][0],
                           "write_kib_avg": (write[k][1] / write[k][0]) if k in write else None}
                       for k in sorted(fetch) if is_conv(k)},
    }
    # what bench.py checks before it quotes this file (a stale profile is silently wrong otherwise): the share of the conv_gemm
    # launches each kernel class takes, and the commit the passes ran on (the GPU box has no .git: passed in by the caller)
    kinds = collections.Counter()
    for k, v in fetch.items():
        if is_conv(k):
            kinds[kernel_kind(k)] += v[0]
    out["launch_share_by_kind"] = {k: v / n for k, v in sorted(kinds.items())}
    out["head"] = sys.argv[4] if len(sys.argv) > 4 else None
    txt = json.dumps(out, indent=1)
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
