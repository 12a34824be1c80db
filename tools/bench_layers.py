#!/usr/bin/env python3
"""GPU microbenchmark of the fused 64-channel layer kernels at the benchmark shape (T=200, 64x64 latent):
    python tools/bench_layers.py   -> us per launch with / without the split-operand (bf16 pipe) projections."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dawn_pytorch_amd.ops import HipOps
from dawn_pytorch_amd.pack import pack_kn, pack_bf3, pack_bf3_temporal_out, rel_pos_bucket

ops = HipOps()
dev = "cuda"
F, HW, win = 200, 4096, 40
torch.manual_seed(0)
x = torch.randn(F * HW, 64, device=dev)
wqkv_kn = torch.randn(64, 768) * 0.125
wqkv, wqkv_s = pack_kn(wqkv_kn).to(dev), pack_bf3(wqkv_kn).to(dev)
wout_kn = torch.randn(256, 64) / 16
wout = pack_kn(wout_kn).to(dev)
wout_sp = pack_bf3_temporal_out(wout_kn).to(dev)
bias = torch.randn(64, device=dev)
pos = torch.arange(F + 2 * win, dtype=torch.float32)
freqs = 10000.0 ** (-torch.arange(0, 32, 2, dtype=torch.float32) / 32)
ang = pos[:, None] * freqs[None, :]
rc, rs = torch.cos(ang).to(dev), torch.sin(ang).to(dev)
band = (torch.randn(2 * win + 1, 8) * 0.1).to(dev)


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for name, s in (("fp32", None), ("split", wqkv_s)):
    t = timeit(lambda: ops.temporal_layer_c64(x, F, HW, 0, F, win, wqkv, wout, rc, rs, band, wqkv_bf3=s))
    print(f"temporal_layer_c64 {name:6s}: {t:8.1f} us")
for flags, name in ((3, "WMODE 2 (projections on the bf16 pipe)"), (4, "WMODE 3 (+ S, P.V, out-proj on the bf16 pipe)"),
                    (4 | 32, "WMODE 3 with the out-projection on fp32 MFMA"), (4 | 16, "WMODE 3 with interleave hints")):
    ops.temporal_flags = flags
    t = timeit(lambda: ops.temporal_layer_c64(x, F, HW, 0, F, win, wqkv, wout, rc, rs, band, wqkv_bf3=wqkv_s, wout_bf3p=wout_sp))
    print(f"temporal_layer_c64 {name}: {t:8.1f} us")
ops.temporal_flags = 0
# T-shard geometry: 200 own frames + 40 halo frames on each side (interior shard), 240 for an edge shard
xe = torch.randn(280 * HW, 64, device=dev)
rc2, rs2 = torch.cos(torch.arange(280 + 2 * win, dtype=torch.float32)[:, None] * freqs[None, :]).to(dev), torch.sin(torch.arange(280 + 2 * win, dtype=torch.float32)[:, None] * freqs[None, :]).to(dev)
t = timeit(lambda: ops.temporal_layer_c64(xe, 280, HW, 40, 200, win, wqkv, wout, rc2, rs2, band))
print(f"temporal_layer_c64 T-shard interior (Fext=280, Fq=200): {t:8.1f} us")
t = timeit(lambda: ops.temporal_layer_c64(xe[:240 * HW], 240, HW, 0, 200, win, wqkv, wout, rc2, rs2, band))
print(f"temporal_layer_c64 T-shard edge     (Fext=240, Fq=200): {t:8.1f} us")
for name, s in (("fp32", None), ("split", wqkv_s)):
    t = timeit(lambda: ops.sla_layer_c64(x, F, HW, wqkv, wout, bias, wqkv_bf3=s))
    print(f"sla_layer_c64      {name:6s}: {t:8.1f} us (context + apply)")
# fused cross-attention branch (Co = 64), Cin = 64 and 128 (two sources)
wq64 = pack_kn(torch.randn(64, 192) * 0.125).to(dev)
wq128 = pack_kn(torch.randn(128, 192) * 0.09).to(dev)
wo = [pack_kn(torch.randn(64, 64) * 0.125).to(dev) for _ in range(3)]
g3 = (torch.randn(3, 64) * 0.2 + 1).to(dev)
q_scale = (torch.rand(3, 8) + 0.5).to(dev)
kvtab = torch.randn(F, 3, 128, device=dev)
nulltab = torch.randn(3, 16, device=dev)
x2 = torch.randn(F * HW, 64, device=dev)
xtab = ops.xattn_tables(kvtab, nulltab, q_scale, wo, 64)            # once per clip in the product
from dawn_pytorch_amd.pack import unpack_kn
for name, s64, s128 in (("fp32 ", None, None), ("split", pack_bf3(unpack_kn(wq64.cpu())).to(dev), pack_bf3(unpack_kn(wq128.cpu())).to(dev))):
    t = timeit(lambda: ops.xattn_layer_c64(x, None, HW, wq64, wo, g3, q_scale, kvtab, nulltab, xtab=xtab, wq_bf3=s64))
    print(f"xattn_layer_c64 Cin=64    {name}: {t:8.1f} us")
    t = timeit(lambda: ops.xattn_layer_c64(x, x2, HW, wq128, wo, g3, q_scale, kvtab, nulltab, xtab=xtab, wq_bf3=s128))
    print(f"xattn_layer_c64 Cin=64+64 {name}: {t:8.1f} us")
