#!/usr/bin/env python3
"""Build container only (minutes of CPU): the oracle's sampler against the reference's 50-step trajectory at T=96
(tests/golden/ddim_T96S50.npz, tools/gen_goldens_ddim.py).  Prints the errors that profiles/r3_parity_errors.md quotes."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import dawn_oracle as O  # noqa: E402
import dawn_pytorch_amd as D  # noqa: E402
from fullsize_cases import DDIM_CASES, KW, build_inputs, ddim_noises  # noqa: E402

torch.set_grad_enabled(False)
for name in sys.argv[1:] or ["C1", "T96S50"]:
    g = np.load(os.path.join(ROOT, "tests", "golden", f"ddim_{name}.npz"))
    T, h, S, keep = DDIM_CASES[name]
    unet = D.DynamicNfUnet3D(default_num_frames=T, **KW, init_seed=0)
    sd = {"denoise_fn." + k: v for k, v in unet.state_dict().items()}
    fea272, cond, x3 = build_inputs(T, h)
    noises = ddim_noises(T, h, S, int(g["ddim_noise_seed"])) + [None]
    trace = []
    out = O.ddim_sample(sd, fea272, cond, x3, noises, S, win=40, trace=trace)
    q = torch.stack([tr["s"][0] for tr in trace]).reshape(-1)
    qref = torch.from_numpy(np.maximum(g["quantiles"], 1.0))
    print(f"{name}: oracle vs reference: thresholds rel {float(((q - qref).abs() / qref).max()):.2e}; "
          + "; ".join(f"x before step {s}: {float((trace[s - 1]['img'][0] - torch.from_numpy(g[f'x_before_step_{s}'])).abs().max()):.2e}" for s in keep)
          + f"; final {float((out[0] - torch.from_numpy(g['out'])).abs().max()):.2e}")
