#!/usr/bin/env python3
"""GPU box: the six resampling convs of the denoiser (Downsample 4x4 / stride 2, Upsample transposed 4x4 as 2x2 phases; gemm1x1_rowacc_kernel MODE 1 / 2)
at the benchmark's shapes, HIP events.   python tools/bench_resample.py [--iters 10] [--policies 0 0x40000000]
(0x40000000: instrumented builds only -- tools/build_timing_lib.sh --, every lane gathers lane 0's pixel: the cost of the gather's scattered line touches)"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dawn_pytorch_amd.ops import HipOps
from dawn_pytorch_amd.pack import pack_kn, pack_bf3, conv_w_kn, deconv_w_kn_phases

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--policies", type=lambda v: int(v, 0), nargs="*", default=[0])
a = ap.parse_args()
ops = HipOps()
dev = "cuda"
F = 200
for (kind, H, C) in (("down", 64, 64), ("down", 32, 128), ("down", 16, 256), ("up", 8, 256), ("up", 16, 128), ("up", 32, 64)):
    g = torch.Generator().manual_seed(1)
    Ci = C
    Co = C if kind == "down" else C // 2 if False else C
    x = torch.randn(F * H * H, Ci, generator=g).to(dev)
    if kind == "down":
        Co = C
        w5 = torch.randn(Co, Ci, 1, 4, 4, generator=g) * (16 * Ci) ** -0.5
        wkn = conv_w_kn(w5)
        w, ws = pack_kn(wkn).to(dev), pack_bf3(wkn).to(dev)
        kw = dict(F=F, Hi=H, Wi=H, Ho=H // 2, Wo=H // 2, KH=4, KW=4, stride=2, pad=1)
        Mout, K = F * (H // 2) ** 2, 16 * Ci
    else:
        Co = C
        w5 = torch.randn(Ci, Co, 1, 4, 4, generator=g) * (4 * Ci) ** -0.5
        ph = deconv_w_kn_phases(w5)
        w = torch.stack([pack_kn(ph[i]) for i in range(4)], 0).to(dev)
        ws = torch.stack([pack_bf3(ph[i]) for i in range(4)], 0).to(dev)
        kw = dict(F=F, Hi=H, Wi=H, Ho=2 * H, Wo=2 * H, KH=2, KW=2, mode=1)
        Mout, K = F * (2 * H) ** 2, 4 * Ci
    bias = torch.randn(Co, generator=g).to(dev)
    for pol in a.policies:
        ops.conv_policy = pol
        for _ in range(3):
            out = ops.conv_gemm(x, w, Co, bias=bias, w_bf3=ws, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            out = ops.conv_gemm(x, w, Co, bias=bias, w_bf3=ws, **kw)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / a.iters * 1e3
        print(f"{kind:4s} {H:2d}x{H:<2d} C={C:3d}  M={Mout:6d} N={Co:3d} K={K:4d}  policy {pol:#x}: {us:7.1f} us  {2.0 * Mout * Co * K / us / 1e6:6.1f} alg TF/s")
