#!/usr/bin/env python3
"""GPU tool: s_memtime phase profile of temporal_layer_c64_kernel (needs the instrumented build:
    hipcc ... -DDAWN_TL_TIMING on temporal_layer.hip, see tools/build_timing_lib.sh).  Prints mean cycles between stamps."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dawn_pytorch_amd.ops import HipOps
from dawn_pytorch_amd.pack import pack_kn, pack_bf3, pack_bf3_temporal_out
ops = HipOps()
dev = "cuda"
F, HW, win = (int(sys.argv[1]) if len(sys.argv) > 1 else 200), 4096, 40
torch.manual_seed(0)
x = torch.randn(F * HW, 64, device=dev)
wqkv_kn = torch.randn(64, 768) * 0.125
wqkv, wqkv_s = pack_kn(wqkv_kn).to(dev), pack_bf3(wqkv_kn).to(dev)
wout_kn = torch.randn(256, 64) / 16
wout = pack_kn(wout_kn).to(dev)
wout_sp = pack_bf3_temporal_out(wout_kn).to(dev)
pos = torch.arange(F + 2 * win, dtype=torch.float32)
freqs = 10000.0 ** (-torch.arange(0, 32, 2, dtype=torch.float32) / 32)
ang = pos[:, None] * freqs[None, :]
rc, rs = torch.cos(ang).to(dev), torch.sin(ang).to(dev)
band = (torch.randn(2 * win + 1, 8) * 0.1).to(dev)
dbg = torch.zeros(512 * 8 * 24, dtype=torch.int64, device=dev)
ops.L.dawn_temporal_set_debug.argtypes = [ctypes.c_void_p]
assert ops.L.dawn_temporal_set_debug(dbg.data_ptr()) == 0
names = ["start", "phase0+setup", "h0 start", "h0 KV proj", "h0 barrier", "h0 Q proj", "h0 S(A)", "h0 S(B)+smA", "h0 PV(A)+smB", "h0 PV(B)", "h0 out",
         "h1 start", "h1 KV proj", "h1 barrier", "h1 Q proj", "h1 S(A)", "h1 S(B)+smA", "h1 PV(A)+smB", "h1 PV(B)", "h1 out", "end (6 more heads + store)"]
for label, s, flags in (
                        ("all-bf16-pipe (WMODE 3, runtime row stride)", wqkv_s, 4 | 64), ("WMODE 3 with interleave hints", wqkv_s, 4 | 64 | 16)):
    ops.temporal_flags = flags
    for _ in range(2):
        dbg.zero_()
        ops.temporal_layer_c64(x, F, HW, 0, F, win, wqkv, wout, rc, rs, band, wqkv_bf3=s, wout_bf3p=wout_sp)
        torch.cuda.synchronize()
    t = dbg.cpu().numpy().reshape(512, 8, 24).astype(np.float64)
    print(f"--- {label}: mean cycles between stamps, per wave (columns = waves 0..7; wave 7 has no query tile)")
    for i in range(20):
        row = []
        for w in range(8):
            a, b = t[:, w, i], t[:, w, i + 1]
            ok = (a != 0) & (b != 0)
            row.append(f"{(b[ok] - a[ok]).mean():8.0f}" if ok.any() else "       -")
        print(f"  {names[i]:14s} -> {names[i+1][:12]:12s}: " + " ".join(row))
