#!/usr/bin/env python3
"""GPU microbenchmark of the split-operand 1x1 GEMMs at the row counts of the deeper levels (M = 12,800 / 51,200):
    python tools/bench_gemm1x1_deep.py [--iters 20] [--policy 0x2B08580D]
Back-to-back launches of one shape (operands warm in L2 / the memory-side cache) and launches separated by a 1 GB fill (cold)."""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dawn_pytorch_amd.ops import HipOps
from dawn_pytorch_amd.pack import pack_kn, pack_bf3

CASES = [  # (M, N, K, LayerNorm prologue)
    (12800, 768, 512, True), (12800, 512, 256, False), (12800, 192, 512, True), (12800, 768, 256, True),
    (12800, 192, 1024, True), (12800, 256, 1024, False), (12800, 192, 256, True), (12800, 256, 256, False),
    (51200, 768, 256, True), (51200, 256, 256, False), (51200, 768, 128, True), (204800, 768, 128, True),
]
ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--first", type=int, default=0, help="only the first N shapes")
ap.add_argument("--policy", type=lambda v: int(v, 0), default=0, help="dawn_conv_desc.policy (0 = shipped; 0x2B08580D = launch-order tiles)")
a = ap.parse_args()
ops = HipOps()
ops.conv_policy = a.policy
dev = "cuda"
junk = torch.empty(256 << 20, device=dev)
for M, N, K, ln in (CASES[:a.first] if a.first else CASES):
    torch.manual_seed(0)
    x = torch.randn(M, K, device=dev)
    w_kn = torch.randn(K, N) * K ** -0.5
    w, ws = pack_kn(w_kn).to(dev), pack_bf3(w_kn).to(dev)
    out = torch.empty(M, N, device=dev)
    F = M // 64
    kw = dict(F=F, Hi=8, Wi=8, KH=1, KW=1, stride=1, pad=0, w_bf3=ws, out=out)
    if ln:
        kw["row_stats"] = (torch.randn(M, device=dev) * 0.1, torch.rand(M, device=dev) + 0.5)
    ops.conv_gemm(x, w, N, **kw)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    e[0].record()
    for _ in range(a.iters):
        ops.conv_gemm(x, w, N, **kw)
    e[1].record()
    torch.cuda.synchronize()
    warm = e[0].elapsed_time(e[1]) * 1e3 / a.iters
    cold = 0.0
    for _ in range(5):
        junk.fill_(1.0)
        e[0].record()
        ops.conv_gemm(x, w, N, **kw)
        e[1].record()
        torch.cuda.synchronize()
        cold += e[0].elapsed_time(e[1]) * 1e3 / 5
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K} ln={int(ln)}: warm {warm:7.1f} us ({fl / warm / 1e6:6.1f} TF/s)   after a 1 GB fill {cold:7.1f} us")
