#!/usr/bin/env python3
"""GPU microbenchmark of the split-operand 1x1 GEMMs (dawn_conv_gemm with w_bf3) at the benchmark's projection shapes.
    python tools/bench_gemm1x1.py [--policy 0x580D,0x2580D] [--iters 20]"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dawn_pytorch_amd.ops import HipOps
from dawn_pytorch_amd.pack import pack_kn, pack_bf3

SHAPES = [  # (M, N, K, row stats, residual)
    (204800, 768, 128, True, False), (51200, 768, 256, True, False), (12800, 768, 512, True, False),
    (204800, 128, 256, False, True), (51200, 256, 256, False, True), (204800, 192, 256, True, False),
    (204800, 192, 128, True, False), (51200, 768, 128, True, False), (51200, 128, 512, False, True),
    (819200, 64, 128, False, True), (204800, 128, 64, False, False),
    (819200, 192, 256, True, False), (51200, 192, 512, True, False), (12800, 192, 1024, True, False), (12800, 192, 512, True, False),
]
ap = argparse.ArgumentParser()
ap.add_argument("--policy", default="0")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--first", type=int, default=0, help="only the first N shapes")
ap.add_argument("--only", type=int, default=-1, help="only shape i of the list")
a = ap.parse_args()
ops = HipOps()
dev = "cuda"
for pol in [int(v, 0) for v in a.policy.split(",")]:
    ops.conv_policy = pol
    for (M, N, K, rs, res) in ([SHAPES[a.only]] if a.only >= 0 else (SHAPES[:a.first] if a.first else SHAPES)):
        torch.manual_seed(0)
        x = torch.randn(M, K, device=dev)
        w_kn = torch.randn(K, N) * K ** -0.5
        kw = dict(F=M // 1024, Hi=32, Wi=32, w_bf3=pack_bf3(w_kn).to(dev))
        if rs:
            kw["row_stats"] = (torch.randn(M, device=dev) * 0.1, torch.rand(M, device=dev) + 0.5)
        if res:
            kw["res"] = torch.randn(M, N, device=dev)
        w = pack_kn(w_kn).to(dev)
        out = torch.empty(M, N, device=dev)
        ops.conv_gemm(x, w, N, out=out, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            ops.conv_gemm(x, w, N, out=out, **kw)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.iters
        print(f"policy {pol:#x} M={M:6d} N={N:3d} K={K:3d}{' ln' if rs else '   '}{' res' if res else '    '}: {us:8.1f} us "
              f"{2.0 * M * N * K / us / 1e6:6.1f} TFLOP/s  {(M * K + M * N * (2 if res else 1)) * 4 / us / 1e6:5.2f} TB/s")
