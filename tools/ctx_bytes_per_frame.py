#!/usr/bin/env python3
"""GPU (weights must live on the device; the sizing itself is a host dry pass): bytes per frame of the C-side evaluator's caller-owned
memory (dawn_workspace_bytes + dawn_clip_bytes) at 256x256, against the Python host's allocator peak (bench.max_clip_frames)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
dev = torch.device("cuda", 0)
unet, diff = bench.build_model(200, 64, 50, dev)
ev = unet.ctx_evaluator()
L, h = ev.L, ev.h
rows = []
for T in (200, 400, 1600, 4800, 6400):
    rows.append((T, int(L.dawn_workspace_bytes(h, T, 64, 64)), int(L.dawn_clip_bytes(h, T, 64, 64))))
    print(f"T={T}: workspace {rows[-1][1] / 1e6:9.1f} MB  clip tables {rows[-1][2] / 1e6:8.1f} MB")
(t1, w1, c1), (t2, w2, c2) = rows[-2], rows[-1]
print(f"per frame: workspace {(w2 - w1) / (t2 - t1) / 1e6:.2f} MB + clip tables {(c2 - c1) / (t2 - t1) / 1e6:.2f} MB")
