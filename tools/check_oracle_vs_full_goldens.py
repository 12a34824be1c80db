#!/usr/bin/env python3
"""Oracle vs the reference-generated full-architecture goldens (tests/golden/full_*.npz) -- minutes of CPU, so it
lives here and not in the `-m "not gpu"` suite (which checks T96 only).  Prints max-abs errors (profiles/r2_parity_errors.md)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import dawn_pytorch_amd as D
from oracle import dawn_oracle as O
from fullsize_cases import CASES, KW, build_inputs
torch.set_grad_enabled(False)
unet = D.DynamicNfUnet3D(default_num_frames=8, **KW, init_seed=0)
sd = {"denoise_fn." + k: v for k, v in unet.state_dict().items()}
for name in (sys.argv[1:] or list(CASES)):
    T, h, tval = CASES[name]
    g = np.load(os.path.join(ROOT, "tests", "golden", f"full_{name}.npz"))
    fea272, cond, x3 = build_inputs(T, h)
    xin = torch.cat((x3, fea272.unsqueeze(2).expand(-1, -1, T, -1, -1)), 1)
    t0 = time.time()
    y = O.unet_forward(sd, xin, torch.tensor([tval]), cond, win=40)
    dt = time.time() - t0
    err = float((y[0][:, torch.from_numpy(g["frames"]).long()] - torch.from_numpy(g["y"])).abs().max())
    print(f"| oracle vs reference, full architecture {name} (T={T}, h={h}) | {err:.3e} | {float(g['y_absmax']):.3f} | oracle {dt:.1f} s, reference {float(g['ref_seconds']):.1f} s on {torch.get_num_threads()} threads |", flush=True)
