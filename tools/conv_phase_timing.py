#!/usr/bin/env python3
"""GPU tool: per-phase s_memtime profile of the split-operand 3x3 conv kernel (instrumented build, ABL bit 3).
    python tools/conv_phase_timing.py [--case l0_3x3_cat]
Prints, averaged over workgroups, the cycles between consecutive stamps (see TSTAMP() in conv_gemm.hip)."""
import argparse, ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dawn_pytorch_amd.ops import HipOps
from dawn_pytorch_amd.pack import pack_kn, pack_bf3

ap = argparse.ArgumentParser()
ap.add_argument("--C0", type=int, default=64)
ap.add_argument("--C1", type=int, default=64)
ap.add_argument("--N", type=int, default=64)
ap.add_argument("--hw", type=int, default=64)
ap.add_argument("--frames", type=int, default=200)
ap.add_argument("--variant", type=lambda v: int(v, 0), default=0x580D | (8 << 16))
a = ap.parse_args()
ops = HipOps()
F, H, W, C0, C1, N = a.frames, a.hw, a.hw, a.C0, a.C1, a.N
rows, K = F * H * W, 9 * (C0 + C1)
torch.manual_seed(0)
x0 = torch.randn(rows, C0, device="cuda")
x1 = torch.randn(rows, C1, device="cuda") if C1 else None
w_kn = torch.randn(K, N) * K ** -0.5
w, ws = pack_kn(w_kn).cuda(), pack_bf3(w_kn).cuda()
dbg = torch.zeros(4096 * 64, dtype=torch.int64, device="cuda")
ops.L.dawn_conv_set_debug.argtypes = [ctypes.c_void_p]
assert ops.L.dawn_conv_set_debug(dbg.data_ptr()) == 0
part = ops.conv_gn_part(rows, N, x0)
for it in range(3):
    ops.conv_policy = a.variant
    dbg.zero_()
    out = ops.conv_gemm(x0, w, N, in1=x1, F=F, Hi=H, Wi=W, KH=3, KW=3, pad=1, w_bf3=ws, gn_part=part)
    torch.cuda.synchronize()
t = dbg.cpu().numpy().reshape(4096, 64)
nb = min(4096, rows // 256 * ((N + 63) // 64))
t = t[:nb]
nst = int((t[0] != 0).sum())
t = t[:, :nst].astype(np.float64)
t0 = t[:, 0].min()
print(f"blocks {nb}, stamps/block {nst}; kernel span {(t[:, -1].max() - t0):.0f} ticks")
d = np.diff(t, axis=1)
print("stamp deltas (mean / p10 / p90 over blocks):")
for i in range(nst - 1):
    print(f"  {i:2d}->{i+1:2d}: {d[:, i].mean():9.0f} {np.percentile(d[:, i], 10):9.0f} {np.percentile(d[:, i], 90):9.0f}")
print(f"block duration mean {(t[:, -1] - t[:, 0]).mean():.0f}; start times (rel) percentiles: "
      + " ".join(f"{np.percentile(t[:, 0] - t0, p):.0f}" for p in (0, 10, 25, 50, 75, 90, 100)))
