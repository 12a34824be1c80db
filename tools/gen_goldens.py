#!/usr/bin/env python3
"""Generate golden vectors by IMPORTING THE REFERENCE in the build container.

Run (build container only; /root/reference does not exist on the GPU box):

    python tools/gen_goldens.py

Writes small ``.npz`` fixtures (data only: inputs, weights, expected outputs) under
``tests/golden/``.  The reference's Python never leaves this container.  Three stub
packages under ``tools/ref_stubs`` stand in for wheels that are absent offline
(SURVEY.md Appendix B): ``einops_exts`` (a map of einops.rearrange), ``torchvision``
(never invoked) and ``rotary_embedding_torch`` (published algorithm restated; this
library boundary is "parity unpinned").
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
OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)

import DM_3.modules.video_flow_diffusion_multiGPU_v0_crema_plus_faceemb_ca_multi_test as MT  # noqa: E402
import DM_3.modules.video_flow_diffusion_multiGPU_v0_crema_plus_faceemb_ca_multi_test_local_opt as MTL  # noqa: E402

torch.set_grad_enabled(False)

TINY = dict(dim=16, cond_dim=24 + 6 + 2, cond_aud=24, cond_pose=6, cond_eye=2, num_frames=12,
            channels=3 + 16, out_grid_dim=2, out_conf_dim=1, dim_mults=(1, 2),
            use_hubert_audio_cond=True, learn_null_cond=False, use_final_activation=False,
            use_deconv=True, padding_mode="zeros", win_width=3)


def np_sd(sd):
    return {k: v.detach().cpu().numpy() for k, v in sd.items()}


def save(name, **arrs):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrs)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.1f} KiB, {len(arrs)} arrays")


def randomize_(module, seed):
    """Default inits leave many gains at 1 / biases at 0; perturb everything so that every
    parameter is observable in the goldens."""
    g = torch.Generator().manual_seed(seed)
    for name, p in module.named_parameters():
        if name.endswith("freqs"):
            continue
        p.add_(torch.randn(p.shape, generator=g) * 0.05)


# ------------------------------------------------------------------ 1. tables
def gen_tables():
    rel = torch.arange(-45, 46)
    bucket = MT.RelativePositionBias._relative_position_bucket(rel, num_buckets=32, max_distance=32)
    diff = MT.GaussianDiffusion(MT.Unet3D(**{**TINY}), image_size=8, num_frames=12, sampling_timesteps=10,
                                use_dynamic_thres=True)
    arrs = {"rel": rel.numpy(), "bucket": bucket.numpy()}
    for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas_cumprod",
              "sqrt_recipm1_alphas_cumprod"):
        arrs["sched_" + k] = getattr(diff, k).numpy()
    for S in (3, 10, 20, 50):
        times = torch.linspace(0., 1000, steps=S + 2)[:-1]
        times = list(reversed(times.int().tolist()))
        arrs[f"times_{S}"] = np.array(times, dtype=np.int64)
        co = []
        for t, tn in zip(times[:-1], times[1:]):
            a, an = diff.alphas_cumprod_prev[t], diff.alphas_cumprod_prev[tn]
            sigma = 1.0 * ((1 - a / an) * (1 - an) / (1 - a)).sqrt()
            c = ((1 - an) - sigma ** 2).sqrt()
            co.append([float(a), float(an), float(sigma), float(c), float(an.sqrt())])
        arrs[f"coef_{S}"] = np.array(co, dtype=np.float64)
    # sinusoidal + full bias matrix for a small T
    arrs["sin_t"] = np.array([0, 19, 980], dtype=np.int64)
    arrs["sin_emb"] = MT.SinusoidalPosEmb(64)(torch.tensor([0, 19, 980])).numpy()
    save("tables.npz", **arrs)


# ------------------------------------------------------------------ 2/3. tiny UNet + per-stage captures
def build_tiny(mod, seed=0):
    torch.manual_seed(seed)
    unet = mod.DynamicNfUnet3D(default_num_frames=12, **TINY)
    randomize_(unet, seed + 1)
    unet.update_num_frames(12)
    unet.eval()
    return unet


def gen_tiny_unet():
    unet = build_tiny(MT)
    unet_l = build_tiny(MTL)
    unet_l.load_state_dict(unet.state_dict())
    g = torch.Generator().manual_seed(123)
    T, h = 12, 8
    # fea/bbox channels are frame-invariant by construction of ddim_sample (MT:1167, 1177)
    x = torch.cat((torch.randn(1, 3, T, h, h, generator=g),
                   torch.randn(1, 16, 1, h, h, generator=g).expand(-1, -1, T, -1, -1)), dim=1).contiguous()
    cond = torch.randn(1, T, 32, generator=g)
    time = torch.tensor([627])

    caps = {}

    def hook(name):
        def f(m, inp, out):
            if name in keep_in:
                caps["in:" + name] = inp[0].detach().clone()
            caps["out:" + name] = out.detach().clone()
        return f

    watch = ["init_conv", "init_temporal_attn", "downs.0.0", "downs.0.0.block1", "downs.0.0.cross_attn_aud",
             "downs.0.1", "downs.0.2", "downs.0.3", "downs.0.4", "downs.1.0", "downs.1.3", "mid_block1",
             "mid_spatial_attn", "mid_temporal_attn", "mid_block2", "ups.0.0", "ups.0.2", "ups.0.4", "ups.1.1",
             "ups.1.3", "final_conv.0", "final_conv", "occlusion_map"]
    keep_in = {"downs.0.0", "downs.0.0.cross_attn_aud", "downs.0.2", "downs.0.3", "downs.0.4",
               "mid_spatial_attn", "ups.0.4", "final_conv.0"}
    mods = dict(unet.named_modules())
    hs = [mods[n].register_forward_hook(hook(n)) for n in watch]
    y = unet.forward_with_cond_scale(x, time, cond=cond, cond_scale=1.0)
    for h_ in hs:
        h_.remove()
    y_l = unet_l.forward_with_cond_scale(x, time, cond=cond, cond_scale=1.0)
    print("tiny MT vs MTL max abs diff", float((y - y_l).abs().max()))
    assert float((y - y_l).abs().max()) < 2e-5
    y_cs = unet.forward_with_cond_scale(x, time, cond=cond, cond_scale=2.5)
    # cross-attention context as seen by the module (captured input[0] is x tokens only)
    arrs = {"x": x.numpy(), "cond": cond.numpy(), "time": time.numpy(), "y": y.numpy(), "y_local": y_l.numpy(),
            "y_cond_scale_2p5": y_cs.numpy(), "win": np.array(3)}
    for k, v in caps.items():
        arrs["cap:" + k] = v.numpy()
    for k, v in np_sd(unet.state_dict()).items():
        arrs["sd:denoise_fn." + k] = v
    save("tiny_unet.npz", **arrs)

    # locality / GroupNorm coupling case: T >> w
    g = torch.Generator().manual_seed(7)
    T2 = 24
    unet.update_num_frames(T2)
    x2 = torch.cat((torch.randn(1, 3, T2, 8, 8, generator=g),
                    torch.randn(1, 16, 1, 8, 8, generator=g).expand(-1, -1, T2, -1, -1)), dim=1).contiguous()
    c2 = torch.randn(1, T2, 32, generator=g)
    y2 = unet.forward_with_cond_scale(x2, torch.tensor([39]), cond=c2, cond_scale=1.0)
    save("tiny_unet_T24.npz", x=x2.numpy(), cond=c2.numpy(), time=np.array([39]), y=y2.numpy())
    return unet


# ------------------------------------------------------------------ 4. DDIM trajectory with injected noise
def gen_ddim(unet):
    S, T, h = 3, 12, 8
    unet.update_num_frames(T)
    diff = MT.DynamicNfGaussianDiffusion(default_num_frames=T, denoise_fn=unet, num_frames=T, image_size=h,
                                         sampling_timesteps=S, timesteps=1000, loss_type='l2',
                                         use_dynamic_thres=True, null_cond_prob=0.1, ddim_sampling_eta=1.0)
    diff.update_num_frames(T)
    diff.eval()
    g = torch.Generator().manual_seed(1234)
    fea = torch.randn(1, 12, h, h, generator=g)
    bbox = torch.randn(1, 4, h, h, generator=g)
    cond = torch.randn(1, T, 32, generator=g)
    x_init = torch.randn(1, 3, T, h, h, generator=g) * 1.5     # push |x0| quantile above 1 at some steps
    noises = [torch.randn(1, 3, T, h, h, generator=g) for _ in range(S)]

    draws = {"i": 0}
    real_randn, real_randn_like, real_quantile = torch.randn, torch.randn_like, torch.quantile
    qs = []

    def fake_randn(*a, **k):
        return x_init.clone()

    def fake_randn_like(t, **k):
        n = noises[draws["i"]]
        draws["i"] += 1
        return n.clone()

    def spy_quantile(*a, **k):
        r = real_quantile(*a, **k)
        qs.append(r.clone())
        return r

    torch.randn, torch.randn_like, torch.quantile = fake_randn, fake_randn_like, spy_quantile
    MT.torch.randn, MT.torch.randn_like = fake_randn, fake_randn_like
    try:
        out = diff.sample(fea, bbox, cond=cond, cond_scale=1.0)
    finally:
        torch.randn, torch.randn_like, torch.quantile = real_randn, real_randn_like, real_quantile
    print("ddim quantiles", [float(q) for q in qs], "noise draws", draws["i"])
    save("ddim_tiny.npz", fea=fea.numpy(), bbox=bbox.numpy(), cond=cond.numpy(), x_init=x_init.numpy(),
         noises=np.stack([n.numpy() for n in noises]), quantiles=np.array([float(q) for q in qs]),
         out=out.numpy(), S=np.array(S))


# ------------------------------------------------------------------ 5. quantile cases
def gen_quantile():
    g = torch.Generator().manual_seed(5)
    cases = {}
    for i, n in enumerate((7, 10, 11, 1000, 12288, 20001)):
        v = torch.randn(2, n, generator=g).abs() * (0.3 if i % 2 else 2.0)
        cases[f"v{i}"] = v.numpy()
        cases[f"q{i}"] = torch.quantile(v, 0.9, dim=-1).numpy()
    v = torch.tensor([[1.0, 1.0, 1.0, 5.0, 5.0, 5.0, 5.0, 5.0, 5.0, 9.0]])
    cases["v_ties"] = v.numpy()
    cases["q_ties"] = torch.quantile(v, 0.9, dim=-1).numpy()
    save("quantile.npz", **cases)


# ------------------------------------------------------------------ 6. FlowDiffusion pre/post (FD)
def gen_fd_prepost():
    """Run the reference's own `sample_one_video` / `generate_bbox_mask` / `Face_loc_Encoder`
    code with the heavyweight collaborators mocked out (LFG generator, diffusion.sample)."""
    for name in ("LFG", "LFG.modules", "LFG.modules.generator", "LFG.modules.bg_motion_predictor",
                 "LFG.modules.region_predictor", "sync_batchnorm", "filter_fourier"):
        m = types.ModuleType(name)
        sys.modules[name] = m
    sys.modules["LFG.modules.generator"].Generator = object
    sys.modules["LFG.modules.bg_motion_predictor"].BGMotionPredictor = object
    sys.modules["LFG.modules.region_predictor"].RegionPredictor = object
    sys.modules["sync_batchnorm"].DataParallelWithCallback = object
    import importlib
    FD = importlib.import_module(
        "DM_3.modules.video_flow_diffusion_model_multiGPU_v0_crema_vgg_floss_plus_faceemb_flow_fast_init_cond_test")
    torch.Tensor.cuda = lambda self, *a, **k: self          # the reference hard-codes .cuda() (FD:193-194)

    torch.manual_seed(3)
    enc = FD.Face_loc_Encoder()
    g = torch.Generator().manual_seed(11)
    B, T, H = 2, 5, 64
    img = torch.rand(B, 3, H, H, generator=g)
    hub = torch.randn(B, T, 1024, generator=g)
    pose = torch.randn(B, 7, T, generator=g)
    eye = torch.rand(B, 2, T, generator=g)
    bbox = torch.tensor([[30., 200., 40., 220., 256., 256.], [10., 100., 20., 90., 128., 128.]]).reshape(B, 6, 1)
    init_pose = torch.randn(B, 7, generator=g)
    init_eye = torch.rand(B, 2, generator=g)

    captured = {}

    class FakeDiffusion:
        def sample(self, fea, bbox_mask, cond=None, batch_size=None, cond_scale=None):
            captured.update(fea=fea.clone(), bbox_mask=bbox_mask.clone(), cond=cond.clone())
            gg = torch.Generator().manual_seed(17)
            captured["pred"] = torch.randn(B, 3, T, H // 4, H // 4, generator=gg)
            return captured["pred"]

    class FakeGen:
        def compute_fea(self, im):
            return im[:, :1, ::4, ::4].repeat(1, 256, 1, 1)

        def forward_with_flow(self, source_image, optical_flow, occlusion_map):
            return {"prediction": source_image, "deformed": source_image}

    fake = types.SimpleNamespace(generator=FakeGen(), diffusion=FakeDiffusion(), face_loc_emb=enc, pose_dim=6,
                                 use_residual_flow=False)
    fake.generate_bbox_mask = types.MethodType(FD.FlowDiffusion.generate_bbox_mask, fake)
    arrs = {}
    for tag, ip, ie in (("given", init_pose, init_eye), ("none", None, None)):
        out = FD.FlowDiffusion.sample_one_video(fake, img.clone(), hub.clone(), pose.clone(), eye.clone(),
                                                bbox.clone(), 1.0, init_pose=ip, init_eye=ie)
        arrs[f"cond_{tag}"] = captured["cond"].numpy()
        arrs[f"bbox_mask_{tag}"] = captured["bbox_mask"].numpy()
        arrs[f"grid_{tag}"] = out["sample_vid_grid"].numpy()
        arrs[f"conf_{tag}"] = out["sample_vid_conf"].numpy()
    arrs["pred"] = captured["pred"].numpy()
    raw = FD.FlowDiffusion.generate_bbox_mask(fake, bbox.clone(), size=H)
    arrs.update(img=img.numpy(), hubert=hub.numpy(), pose=pose.numpy(), eye=eye.numpy(), bbox=bbox.numpy(),
                init_pose=init_pose.numpy(), init_eye=init_eye.numpy(), raw_mask=raw.numpy())
    for k, v in np_sd(enc.state_dict()).items():
        arrs["enc:" + k] = v
    save("fd_prepost.npz", **arrs)


if __name__ == "__main__":
    gen_tables()
    unet = gen_tiny_unet()
    gen_ddim(unet)
    gen_quantile()
    gen_fd_prepost()
