#!/usr/bin/env python3
"""GPU: is the fused temporal layer (WMODE 3) limited by the chip's power budget like the split convs?  The same launch on N(0,1)
activations / weights and on zeros (same instruction stream: LayerNorm of a zero row is zero, every product is zero), alternating."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dawn_pytorch_amd.ops import HipOps
from dawn_pytorch_amd.pack import pack_kn, pack_bf3, pack_bf3_temporal_out

ops = HipOps()
dev = "cuda"
F, HW, win = 200, 4096, 40
torch.manual_seed(0)
pos = torch.arange(F + 2 * win, dtype=torch.float32)
freqs = 10000.0 ** (-torch.arange(0, 32, 2, dtype=torch.float32) / 32)
ang = pos[:, None] * freqs[None, :]
rc, rs = torch.cos(ang).to(dev), torch.sin(ang).to(dev)


def setup(zero):
    x = torch.zeros(F * HW, 64, device=dev) if zero else torch.randn(F * HW, 64, device=dev)
    wq = torch.zeros(64, 768) if zero else torch.randn(64, 768) * 0.125
    wo = torch.zeros(256, 64) if zero else torch.randn(256, 64) / 16
    band = (torch.zeros(2 * win + 1, 8) if zero else torch.randn(2 * win + 1, 8) * 0.1).to(dev)
    return x, pack_kn(wq).to(dev), pack_bf3(wq).to(dev), pack_kn(wo).to(dev), pack_bf3_temporal_out(wo).to(dev), band


def timeit(a, n=20):
    x, wqkv, wqkv_s, wout, wout_sp, band = a
    f = lambda: ops.temporal_layer_c64(x, F, HW, 0, F, win, wqkv, wout, rc, rs, band, wqkv_bf3=wqkv_s, wout_bf3p=wout_sp)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


rnd, zer = setup(False), setup(True)
for r in range(3):
    print(f"round {r}: N(0,1) data {timeit(rnd):8.1f} us   zeros {timeit(zer):8.1f} us")
