#!/usr/bin/env python3
"""GPU microbenchmark of the fused 64-channel spatial-linear-attention layer (dawn_sla_layer_c64, split-operand form) at the benchmark shape
(200 frames of 64 x 64 pixels): python tools/bench_sla_layer.py [--frames 200] [--hw 4096]"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dawn_pytorch_amd.ops import HipOps
from dawn_pytorch_amd.pack import pack_kn, pack_bf3
ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=200)
ap.add_argument("--hw", type=int, default=4096)
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
ops = HipOps()
dev = "cuda"
torch.manual_seed(0)
F, HW = a.frames, a.hw
x = torch.randn(F * HW, 64, device=dev)
wqkv_kn = torch.randn(64, 768) * 0.125
wout_kn = torch.randn(256, 64) / 16
wqkv, wqkv_s, wout = pack_kn(wqkv_kn).to(dev), pack_bf3(wqkv_kn).to(dev), pack_kn(wout_kn).to(dev)
bias = torch.randn(64, device=dev) * 0.1
out = torch.empty_like(x)
for _ in range(2):
    ops.sla_layer_c64(x, F, HW, wqkv, wout, bias, wqkv_bf3=wqkv_s, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters):
    ops.sla_layer_c64(x, F, HW, wqkv, wout, bias, wqkv_bf3=wqkv_s, out=out)
e1.record()
torch.cuda.synchronize()
print(f"sla_layer_c64 F={F} HW={HW}: {e0.elapsed_time(e1) * 1e3 / a.iters:8.1f} us per layer (context + merge + apply); checksum {float(out.double().sum()):.6f}")
