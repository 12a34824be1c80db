#!/usr/bin/env python3
"""GPU tool: s_memtime phase profile of the window-tiled fused temporal layer (csrc/temporal_layer16.hip).  Needs the instrumented
library: tools/build_tl16_debug_lib.sh -DDAWN_TL_TIMING, then DAWN_HIP_LIB=tools/ubench/libdawn_hip_tl16debug.bin.
Prints mean cycles between stamps per wave (stamps cost cycles themselves: read the proportions, not the totals)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dawn_pytorch_amd.ops import HipOps
from dawn_pytorch_amd.pack import pack_kn, pack_bf3, pack_bf3_temporal_out
ops = HipOps()
dev = "cuda"
F, HW, win = (int(sys.argv[1]) if len(sys.argv) > 1 else 200), 4096, 40
NW = 8
torch.manual_seed(0)
x = torch.randn(F * HW, 64, device=dev)
wqkv_kn = torch.randn(64, 768) * 0.125
wqkv, wqkv_s = pack_kn(wqkv_kn).to(dev), pack_bf3(wqkv_kn).to(dev)
wout_kn = torch.randn(256, 64) / 16
wout, wout_sp = pack_kn(wout_kn).to(dev), pack_bf3_temporal_out(wout_kn).to(dev)
ang = torch.arange(F, dtype=torch.float32)[:, None] * (10000.0 ** (-torch.arange(0, 32, 2, dtype=torch.float32) / 32))[None, :]
rc, rs = torch.cos(ang).to(dev), torch.sin(ang).to(dev)
band = (torch.randn(2 * win + 1, 8) * 0.1).to(dev)
dbg = torch.zeros(512 * NW * 20, dtype=torch.int64, device=dev)
ops.L.dawn_temporal16_set_debug.argtypes = [ctypes.c_void_p]
assert ops.L.dawn_temporal16_set_debug(dbg.data_ptr()) == 0
names = ["start", "phase 0", "h0 start", "h0 Q proj", "h0 K/V group", "h0 barrier A", "h0 S+softmax (tile 0)", "h0 P.V (tile 0)", "h0 out-proj (tile 0)",
         "h0 phase B end", "h1 start", "h1 Q proj", "h1 K/V group", "h1 barrier A", "h1 S+softmax (tile 0)", "h1 P.V (tile 0)", "h1 out-proj (tile 0)",
         "h1 phase B end", "end (6 more heads + store)"]
ops.temporal_flags = 5
for _ in range(2):
    dbg.zero_()
    ops.temporal_layer_c64(x, F, HW, 0, F, win, wqkv, wout, rc, rs, band, wqkv_bf3=wqkv_s, wout_bf3p=wout_sp)
    torch.cuda.synchronize()
t = dbg.cpu().numpy().reshape(512, NW, 20).astype(np.float64)
print(f"--- window-tiled layer, F = {F}: mean cycles between stamps, per wave (columns = waves 0..{NW - 1}; waves w, w + 4 share a SIMD)")
for i in range(len(names) - 1):
    row = []
    for w in range(NW):
        a, b = t[:, w, i], t[:, w, i + 1]
        ok = (a != 0) & (b != 0)
        row.append(f"{(b[ok] - a[ok]).mean():8.0f}" if ok.any() else "       -")
    print(f"  {names[i]:22s} -> {names[i+1][:20]:20s}: " + " ".join(row))
tot = []
for w in range(NW):
    a = t[:, w, 0]
    b = t[:, w, :].max(axis=1)
    tot.append(f"{(b - a).mean():8.0f}")
print(f"  {'start -> end':46s}: " + " ".join(tot))
