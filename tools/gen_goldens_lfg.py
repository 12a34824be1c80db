#!/usr/bin/env python3
"""Golden vectors for the LFG flow decode, made by IMPORTING THE REFERENCE's `Generator`
(LFG/modules/generator.py) in the build container and running ITS per-frame loop (FD:372-385).

    python tools/gen_goldens_lfg.py        # writes tests/golden/lfg_tiny.npz (data only)

Import stubs (no arithmetic, never invoked on this path): tools/ref_stubs/skimage (util.py imports
`skimage.draw.disk` for a visualisation helper).  The generator is a narrow instance of the shipped
architecture (config/hdtf256.yaml: block_expansion 64, max_features 512, 2 down blocks, 6 bottleneck blocks,
skips) -- same topology, 16/32/64 channels, 2 bottleneck blocks -- with every parameter AND every BatchNorm
running statistic randomised so that each one is observable in the outputs.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("DAWN_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(ROOT, "tools", "ref_stubs"))
sys.path.insert(0, REF)

from LFG.modules.generator import Generator  # noqa: E402

torch.set_grad_enabled(False)


def main():
    torch.manual_seed(7)
    gen = Generator(num_channels=3, num_regions=10, block_expansion=16, max_features=64, num_down_blocks=2,
                    num_bottleneck_blocks=2, skips=True)
    g = torch.Generator().manual_seed(11)
    for name, buf in gen.named_buffers():
        if name.endswith("running_mean"):
            buf.copy_(torch.randn(buf.shape, generator=g) * 0.2)
        elif name.endswith("running_var"):
            buf.copy_(torch.rand(buf.shape, generator=g) * 1.5 + 0.25)
    for name, p in gen.named_parameters():
        p.add_(torch.randn(p.shape, generator=g) * 0.05)
    gen.eval()

    B, T, H, h = 1, 5, 32, 8
    img = torch.rand(B, 3, H, H, generator=g)
    # flow = identity grid + perturbation, with some samples pushed outside [-1, 1] (zero padding is exercised)
    lin = (torch.arange(h, dtype=torch.float32) + 0.5) / h * 2 - 1
    ident = torch.stack(torch.meshgrid(lin, lin, indexing="xy"), 0)                        # (2,h,h): (x, y)
    grid = ident.view(1, 2, 1, h, h) + torch.randn(B, 2, T, h, h, generator=g) * 0.25
    grid[:, :, 1] *= 1.6
    conf = torch.rand(B, 1, T, h, h, generator=g)

    fea = gen.compute_fea(img)
    outs, warps = [], []
    for idx in range(T):                                                                  # FD:375-383
        o = gen.forward_with_flow(source_image=img, optical_flow=grid[:, :, idx].permute(0, 2, 3, 1),
                                  occlusion_map=conf[:, :, idx])
        outs.append(o["prediction"])
        warps.append(o["deformed"])
    out_vid = torch.stack(outs, dim=2)
    warped_vid = torch.stack(warps, dim=2)

    arrs = {"sd/" + k: v.numpy() for k, v in gen.state_dict().items() if not k.endswith("num_batches_tracked")}
    arrs.update(img=img.numpy(), grid=grid.numpy(), conf=conf.numpy(), fea=fea.numpy(),
                sample_out_vid=out_vid.numpy(), sample_warped_vid=warped_vid.numpy())
    path = os.path.join(ROOT, "tests", "golden", "lfg_tiny.npz")
    np.savez_compressed(path, **arrs)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.1f} KiB, {len(arrs)} arrays; "
          f"out range [{out_vid.min():.3f}, {out_vid.max():.3f}]")


if __name__ == "__main__":
    main()
