#!/usr/bin/env python3
"""Multi-step DDIM trajectories produced by RUNNING THE REFERENCE sampler in the build container.

    python tools/gen_goldens_ddim.py [case ...]        # cases: C1 T96S50 (default: all)

The reference `DynamicNfGaussianDiffusion.sample` (MT:1137-1208: `ddim_sample`, eta = 1, dynamic thresholding at the
0.9 quantile) drives the reference `DynamicNfUnet3D` at the shipped 49.9 M-parameter architecture:

    C1     : T=16, h=32, S=10   = BASELINE configs[0]'s exact workload (128x128, 16 frames, 10 DDIM steps)
    T96S50 : T=96, h=32, S=50   = the benchmark's step count at a clip length where the attention window cuts

Weights = the build's deterministic name-keyed initialisation (`Unet3D(init_seed=0)`, rebuilt from the seed on the GPU
box; fp64 checksum in the fixture); inputs = `fullsize_cases.build_inputs`; the initial latent (torch.randn of
MT:1166) is the seeded `x3`, and the per-step noise (torch.randn_like, MT:1201) is injected from ONE seeded CPU
generator that the tests re-create (`ddim_noise_seed`), so the fixture holds outputs only: the final sample, the latent
after a few intermediate steps, and the dynamic-threshold quantile of EVERY step (torch.quantile, MT:1186-1190).
Data only; the reference's Python never leaves this container.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("DAWN_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(ROOT, "tools", "ref_stubs"))
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
OUT = os.path.join(ROOT, "tests", "golden")

import DM_3.modules.video_flow_diffusion_multiGPU_v0_crema_plus_faceemb_ca_multi_test as MT  # noqa: E402
import dawn_pytorch_amd as D  # noqa: E402
from fullsize_cases import DDIM_CASES, KW, build_inputs, checksum, ddim_noises  # noqa: E402

torch.set_grad_enabled(False)


def main(which):
    ours = D.DynamicNfUnet3D(default_num_frames=8, **KW, init_seed=0)
    sd = ours.state_dict()
    wsum = checksum(sd.values())
    for name in which:
        T, h, S, keep = DDIM_CASES[name]
        fea272, cond, x3 = build_inputs(T, h)
        u = MT.DynamicNfUnet3D(default_num_frames=T, **KW)
        u.update_num_frames(T)
        u.load_state_dict(sd, strict=True)
        u.eval()
        diff = MT.DynamicNfGaussianDiffusion(default_num_frames=T, denoise_fn=u, num_frames=T, image_size=h,
                                             sampling_timesteps=S, timesteps=1000, loss_type='l2', use_dynamic_thres=True,
                                             null_cond_prob=0.1, ddim_sampling_eta=1.0)
        diff.update_num_frames(T)
        diff.eval()
        noises = ddim_noises(T, h, S)
        state = {"n": 0}
        rr, rl, tq = torch.randn, torch.randn_like, torch.quantile
        qs, xs = [], {}
        torch.randn = lambda *a, **k: x3.clone()

        def frl(t, **k):
            n = noises[state["n"]]
            state["n"] += 1
            return n.clone()

        def fq(*a, **k):
            r = tq(*a, **k)
            qs.append(r.reshape(-1).clone())
            return r

        # the latent entering step s is the `x` argument of the UNet call of step s: wrap forward_with_cond_scale
        calls = {"n": 0}
        fwcs = u.forward_with_cond_scale

        def wrapped(x, *a, **k):
            s = calls["n"]
            calls["n"] += 1
            if s in keep:
                xs[s] = x[0, :3].clone()     # the latent BEFORE step s (= after step s-1)
            return fwcs(x, *a, **k)
        u.forward_with_cond_scale = wrapped
        torch.randn_like, torch.quantile = frl, fq
        t0 = time.time()
        try:
            out = diff.sample(fea272[:, :256], fea272[:, 256:], cond=cond, cond_scale=1.0)
        finally:
            torch.randn, torch.randn_like, torch.quantile = rr, rl, tq
            u.forward_with_cond_scale = fwcs
        dt = time.time() - t0
        assert len(qs) == S and state["n"] == S - 1, (len(qs), state["n"])
        arrs = dict(T=T, h=h, S=S, out=out[0].numpy(), quantiles=torch.cat(qs).numpy(), ddim_noise_seed=1234,
                    weights_checksum=wsum, inputs_checksum=checksum([fea272, cond, x3]), ref_seconds=dt,
                    keep=np.asarray(sorted(xs)), **{f"x_before_step_{s}": xs[s].numpy() for s in xs})
        print(f"{name}: reference DDIM T={T} h={h} S={S}: {dt:.1f} s on {torch.get_num_threads()} threads; "
              f"quantiles {torch.cat(qs)[:4].tolist()} ... {torch.cat(qs)[-3:].tolist()}; max|out| = {float(out.abs().max()):.4f}")
        path = os.path.join(OUT, f"ddim_{name}.npz")
        np.savez_compressed(path, **arrs)
        print(f"wrote {path}: {os.path.getsize(path) / 1e6:.2f} MB")


if __name__ == "__main__":
    main(sys.argv[1:] or list(DDIM_CASES))
