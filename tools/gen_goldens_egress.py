#!/usr/bin/env python3
"""Golden vectors for the frame egress produced by CALLING THE REFERENCE's own
`VideoGenerator._process_output_frame` (unified_video_generator.py:533-548) in the build container.

UVG imports wheels that are absent offline (cv2, soundfile, pydub, onnxruntime, a Cython NMS module of 3DDFA).  None
of them does arithmetic on this path: import stubs under tools/ref_stubs (cv2.cvtColor(RGB2BGR) = channel permutation)
and empty stand-ins for the 3DDFA front-end modules make the module importable; the arithmetic that is pinned
(`+ mean/255` in float64 rounded into the float32 frame, clip, *255, astype(uint8)) is the reference's own numpy code.

    python tools/gen_goldens_egress.py   ->  tests/golden/frames_u8.npz
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("DAWN_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(ROOT, "tools", "ref_stubs"))
sys.path.insert(0, REF)
for name, attrs in {"extract_init_states": [], "extract_init_states.FaceBoxes": [],
                    "extract_init_states.FaceBoxes.FaceBoxes_ONNX": ["FaceBoxes_ONNX"],
                    "extract_init_states.TDDFA_ONNX": ["TDDFA_ONNX"], "extract_init_states.utils": [],
                    "extract_init_states.utils.pose": ["get_pose"],
                    "extract_init_states.utils.functions": ["calculate_eye", "calculate_bbox"],
                    "transformers": ["AutoProcessor", "HubertModel"],   # HuBERT stage: not on this path (the real package
                                                                        # mis-detects the torchvision stub)
                    "PBnet": [], "PBnet.src": [], "PBnet.src.models": [], "PBnet.src.models.get_model": ["get_model"]}.items():
    m = types.ModuleType(name)          # 3DDFA / PBnet front-end stages: not on this path, never called
    m.__path__ = []
    for a in attrs:
        setattr(m, a, None)
    sys.modules[name] = m
import unified_video_generator as UVG  # noqa: E402

fn = UVG.VideoGenerator._process_output_frame      # uses no instance state

g = torch.Generator().manual_seed(7)
B, H, W = 6, 24, 40
x = torch.rand(B, 3, H, W, generator=g) * 1.3 - 0.15            # values below 0 and above 1 -> clip
k = torch.arange(0, 256, dtype=torch.float32) / 255.0            # exact k/255 and their float neighbours (truncation edges)
edge = torch.cat([k, torch.nextafter(k, torch.tensor(2.0)), torch.nextafter(k, torch.tensor(-1.0))])
x[0].view(-1)[:edge.numel()] = edge
x[1].view(-1)[:4] = torch.tensor([float("-0.0"), 1.0, 0.999999, 1e-8])
means = [(0.0, 0.0, 0.0), (104.0, 117.0, 123.0), (-3.5, 0.25, 7.0)]
arrs = {"x": x.numpy()}
for mi, mean in enumerate(means):
    out = np.stack([fn(None, x, mean=mean, index=i) for i in range(B)])      # (B,H,W,3) uint8, BGR
    arrs[f"mean{mi}"] = np.asarray(mean, dtype=np.float64)
    arrs[f"bgr{mi}"] = out
    print(f"mean {mean}: {out.shape} {out.dtype}, histogram corners {np.bincount(out.ravel(), minlength=256)[[0, 255]]}")
path = os.path.join(ROOT, "tests", "golden", "frames_u8.npz")
np.savez_compressed(path, **arrs)
print(f"wrote {path}: {os.path.getsize(path) / 1024:.1f} KiB")
