#!/usr/bin/env python3
"""GPU microbenchmark of dawn_conv_gemm on the benchmark's dominant GEMM shapes (C3: T=200, 64x64 latent).
    python tools/bench_conv.py [--cases l0_3x3,l3_3x3,...] [--iters 10]
Prints us/launch and TFLOP/s per case (HIP events around the launch loop).  Used under rocprofv3 --pmc too."""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dawn_pytorch_amd.ops import HipOps
from dawn_pytorch_amd.pack import pack_kn, pack_bf3

CASES = {   # name: (F, H, W, C0, C1, N, k, stride, pad, rowstats)
    "l0_3x3": (200, 64, 64, 64, 0, 64, 3, 1, 1, False),
    "l0_3x3_cat": (200, 64, 64, 64, 64, 64, 3, 1, 1, False),
    "l0_3x3_k2304": (200, 64, 64, 128, 128, 64, 3, 1, 1, False),
    "l0_3x3_k288": (200, 64, 64, 32, 0, 64, 3, 1, 1, False),
    "l1_3x3": (200, 32, 32, 128, 0, 128, 3, 1, 1, False),
    "l2_3x3": (200, 16, 16, 256, 0, 256, 3, 1, 1, False),
    "l3_3x3": (200, 8, 8, 512, 0, 512, 3, 1, 1, False),
    "l0_qkv": (200, 64, 64, 64, 0, 768, 1, 1, 0, True),
    "l0_out": (200, 64, 64, 256, 0, 64, 1, 1, 0, False),
    "l0_xq": (200, 64, 64, 64, 0, 192, 1, 1, 0, True),
    "l0_xo": (200, 64, 64, 64, 0, 64, 1, 1, 0, False),
    "l3_qkv": (200, 8, 8, 512, 0, 768, 1, 1, 0, True),
}
ap = argparse.ArgumentParser()
ap.add_argument("--cases", default=",".join(CASES))
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--variants", default="7")
ap.add_argument("--data", default="randn", choices=["randn", "zero", "const"], help="operand values (the chip clocks to its power budget: zeros run faster than random data)")
ap.add_argument("--gn", action="store_true", help="emit GroupNorm partial sums from the conv epilogue (as the ResBlock convs do)")
a = ap.parse_args()
ops = HipOps()
dev = "cuda"
import itertools
for variant, name in itertools.product([int(v, 0) for v in a.variants.split(",")], a.cases.split(",")):
    ops.conv_policy = variant
    F, H, W, C0, C1, N, k, st, pad, rs = CASES[name]
    rows = F * H * W
    torch.manual_seed(0)
    x0 = torch.randn(rows, C0, device=dev)
    x1 = torch.randn(rows, C1, device=dev) if C1 else None
    K = k * k * (C0 + C1)
    torch.manual_seed(0)
    w_kn = torch.randn(K, N) * K ** -0.5
    if a.data != "randn":
        fill = 0.0 if a.data == "zero" else 0.37
        x0.fill_(fill)
        if x1 is not None:
            x1.fill_(fill)
        w_kn.fill_(fill)
    w = pack_kn(w_kn).to(dev)
    b = torch.randn(N, device=dev)
    kw = dict(F=F, Hi=H, Wi=W, KH=k, KW=k, stride=st, pad=pad, in1=x1, bias=b)
    if k == 3 and (variant & 0x1000):
        kw["w_bf3"] = pack_bf3(w_kn).to(dev)
    if rs:
        kw["row_stats"] = (torch.randn(rows, device=dev) * 0.1, torch.rand(rows, device=dev) + 0.5)
    out = torch.empty(rows, N, device=dev)
    if a.gn and k == 3:
        kw["gn_part"] = ops.conv_gn_part(rows, N, x0)
    ops.conv_gemm(x0, w, N, out=out, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        ops.conv_gemm(x0, w, N, out=out, **kw)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / a.iters
    fl = 2.0 * rows * N * K
    ref_key = (name,)
    refs = globals().setdefault("_refs", {})
    diff = float((out - refs[ref_key]).abs().max()) if ref_key in refs else 0.0
    refs.setdefault(ref_key, out.clone())
    print(f"maxdiff_vs_first_variant={diff:.2e} ", end="")
    print(f"v{variant} {name:12s} M={rows} N={N} K={K}: {us:9.1f} us  {fl / us / 1e6:7.1f} TFLOP/s  "
          f"({(rows * (C0 + C1) + rows * N) * 4 / us / 1e6:6.2f} TB/s min-traffic)")
