#!/usr/bin/env python3
"""GPU: the max-clip-length estimate of bench.py alone (peak allocator bytes per frame of one DDIM step on the long-clip path)."""
import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
dev = torch.device("cuda", 0)
unet, diff = bench.build_model(200, 64, 50, dev)
r = bench.max_clip_frames(unet, diff, 64, dev, 1)
print(json.dumps({k: r[k] for k in ("per_gpu", "bytes_per_frame", "fixed_bytes", "probes")}))
