#!/usr/bin/env python3
"""Golden vectors for SURVEY 8(f) N4 produced by RUNNING THE REFERENCE in the build container:

    python tools/gen_goldens_pbnet.py        -> tests/golden/pbnet_tiny.npz

The reference's own `get_model(parameters)` (PBnet/src/models/get_model.py:17-34) builds CVAE(encoder, decoder) for
  pose : archiname transformerreemb6, pos_dim 6, eye_dim 0      (UVG:80-84)
  blink: archiname transformerreemb5, pos_dim 0, eye_dim 2      (UVG:88-92; reemb6 forces eye_dim = 0, so its blink decoder would
                                                                 have no input -- reemb5 is the same graph with eye_dim honoured)
at REDUCED widths (audio_dim 48, audio latent 24, ff 96, 2 layers, 4 heads of 32; pose latent 64 as shipped) so that the fixture
stays small; every parameter (incl. LayerNorm gains / biases and the relative-position embeddings) is randomised, `.eval()` as
UVG:109-110 does, and `model.generate(init, audio, durations, fact=1)` (cae.py:112-175) runs with the latent z INJECTED
(torch.randn patched) at T = 20, 130 and 210 (the eval-mode window of RelativePositionBias -- +-100 frames in reemb6, +-200 in
reemb5 -- cuts at the longer ones).  Import stubs without arithmetic: tools/ref_stubs (einops_exts, rotary_embedding_torch --
the unpinned library boundary of SURVEY 8c C2).  Data only; the reference's Python never leaves this container."""
import contextlib
import io
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("DAWN_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(ROOT, "tools", "ref_stubs"))
sys.path.insert(0, os.path.join(REF, "PBnet"))
sys.path.insert(0, REF)
from src.models.get_model import get_model  # noqa: E402

torch.set_grad_enabled(False)
BASE = dict(modeltype="cvae", device="cpu", lambdas={"rc": 1.0, "kl": 1e-3}, latent_dim=24, num_frames=32, audio_dim=48,
            pose_latent_dim=64, audio_latent_dim=24, ff_size=96, num_layers=2, num_heads=4, dropout=0.1, activation="gelu")
MODELS = {"pose": dict(archiname="transformerreemb6", pos_dim=6, eye_dim=0), "blink": dict(archiname="transformerreemb5", pos_dim=0, eye_dim=2)}
CASES = [("T20", 20, 20), ("T130", 130, 130), ("T210", 210, 210)]     # (durations == T: lengths_to_mask sizes the mask by max(durations), cae.py:88-94)

arrs = {}
g = torch.Generator().manual_seed(2024)
for name, kw in MODELS.items():
    with contextlib.redirect_stdout(io.StringIO()):
        model = get_model({**BASE, **kw}).eval()
    for k, p in model.decoder.state_dict().items():
        if "rotary_emb.freqs" in k:
            continue
        if k.endswith("norm.gamma") or k.endswith(".weight") and p.dim() == 1:
            p.copy_(1.0 + 0.3 * torch.randn(p.shape, generator=g))
        elif p.dim() == 1:
            p.copy_(0.3 * torch.randn(p.shape, generator=g))
        elif "relative_attention_bias" in k:
            p.copy_(2.0 * torch.randn(p.shape, generator=g))
        else:
            p.copy_(torch.randn(p.shape, generator=g) * (1.5 / p.shape[1] ** 0.5))
    for k, p in model.decoder.state_dict().items():
        if "sequence_pos_encoder" not in k:          # (a 5000 x 64 sinusoid table the decoder's forward never applies)
            arrs[f"sd:{name}:{k}"] = p.numpy().copy()
    din = kw["pos_dim"] + kw["eye_dim"]
    for cname, T, dur in CASES:
        init = torch.rand(1, 1, din, generator=g)
        audio = torch.randn(1, T, BASE["audio_dim"], generator=g)
        z = torch.randn(T, 1, BASE["audio_latent_dim"], generator=g)
        rr = torch.randn
        torch.randn = lambda *a, **k: z.clone()
        try:
            with contextlib.redirect_stdout(io.StringIO()):      # (RelativePositionBias prints 'eval!' in eval mode)
                out = model.generate(init, audio, torch.tensor([dur]), fact=1)["output"]
        finally:
            torch.randn = rr
        arrs.update({f"{name}:{cname}:init": init.numpy(), f"{name}:{cname}:audio": audio.numpy(), f"{name}:{cname}:z": z.numpy(),
                     f"{name}:{cname}:dur": np.asarray([dur]), f"{name}:{cname}:out": out.numpy()})
        print(f"{name} {cname}: out {tuple(out.shape)} max|out| {float(out.abs().max()):.3f}")
arrs["heads"] = np.asarray(BASE["num_heads"])
path = os.path.join(ROOT, "tests", "golden", "pbnet_tiny.npz")
np.savez_compressed(path, **arrs)
print(f"wrote {path}: {os.path.getsize(path) / 1e3:.0f} KB")
