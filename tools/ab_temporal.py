#!/usr/bin/env python3
"""GPU tool: interleaved A/B of temporal-layer kernel variants (flags) in ONE process -- run-to-run / box-to-box noise on the
shared pool is +-3 %, so variants are alternated and the per-variant minimum and median over the repeats are printed.
    python tools/ab_temporal.py 4 4|16 4|128 ..."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dawn_pytorch_amd.ops import HipOps
from dawn_pytorch_amd.pack import pack_kn, pack_bf3, pack_bf3_temporal_out
ops = HipOps()
dev = "cuda"
F, HW, win = 200, 4096, 40
torch.manual_seed(0)
x = torch.randn(F * HW, 64, device=dev)
wqkv_kn, wout_kn = torch.randn(64, 768) * 0.125, torch.randn(256, 64) / 16
wqkv, wqkv_s = pack_kn(wqkv_kn).to(dev), pack_bf3(wqkv_kn).to(dev)
wout, wout_sp = pack_kn(wout_kn).to(dev), pack_bf3_temporal_out(wout_kn).to(dev)
freqs = 10000.0 ** (-torch.arange(0, 32, 2, dtype=torch.float32) / 32)
ang = torch.arange(F + 2 * win, dtype=torch.float32)[:, None] * freqs[None, :]
rc, rs = torch.cos(ang).to(dev), torch.sin(ang).to(dev)
band = (torch.randn(2 * win + 1, 8) * 0.1).to(dev)
variants = [eval(a) for a in sys.argv[1:]] or [4, 4 | 16]
out = torch.empty(F * HW, 64, device=dev)


def run(flags, n=10):
    ops.temporal_flags = flags
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        ops.temporal_layer_c64(x, F, HW, 0, F, win, wqkv, wout, rc, rs, band, wqkv_bf3=wqkv_s, wout_bf3p=wout_sp, out=out)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for v in variants:
    run(v, 3)
res = {v: [] for v in variants}
for rep in range(8):
    for v in variants:
        res[v].append(run(v))
for v in variants:
    a = np.array(res[v])
    print(f"flags {v:4d}: min {a.min():8.1f} us   median {np.median(a):8.1f} us   max {a.max():8.1f} us")
