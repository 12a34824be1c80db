"""-m gpu: the C-side evaluator (dawn_ctx_create / dawn_clip_prepare / dawn_unet_forward /
dawn_sampler_run, csrc/dawn_ctx.hip) against the Python-orchestrated path -- same kernels, same arguments, so the outputs
must be BIT-IDENTICAL -- and against the reference goldens."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from test_hip_end2end import TINY_KW, log, tiny_unet
import dawn_pytorch_amd as D
from dawn_pytorch_amd.ctx import CtxEvaluator, OPT_OVERLAP, OPT_PROFILE
from dawn_pytorch_amd.sampler import ddim_step_scalars
from dawn_pytorch_amd.unet_forward import unet_forward

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def test_ctx_tiny_forward_golden_and_bit_identical(tiny):
    g, sd = tiny
    unet = tiny_unet(sd)
    ops, P = unet._ops(), unet.packed()
    x = T(g["x"]).cuda()
    cond = T(g["cond"]).cuda()
    fea272 = x[0, 3:, 0].contiguous()
    cs = unet.build_clip(fea272, cond[0].contiguous())
    want_py = unet_forward(ops, P, cs, x[0, :3].contiguous(), int(g["time"][0]))
    ev = CtxEvaluator(P)
    for overlap in (1, 0):
        ev.set_option(OPT_OVERLAP, overlap)
        clip = ev.prepare_clip(fea272, cond[0].contiguous(), cs.rcos, cs.rsin)
        got = ev.forward(clip, x[0, :3].contiguous(), float(g["time"][0]))
        assert torch.equal(got, want_py), float((got - want_py).abs().max())
    assert log("ctx_tiny_unet", got[None], T(g["y"])) < 2e-5
    # rotary tables computed by the library itself (fp64 cos/sin rounded once): within round-off of the torch tables
    clip2 = ev.prepare_clip(fea272, cond[0].contiguous())
    got2 = ev.forward(clip2, x[0, :3].contiguous(), float(g["time"][0]))
    assert log("ctx_tiny_unet_own_rotary", got2[None], T(g["y"])) < 2e-5


@pytest.mark.parametrize("Tn,h", [(16, 32), (5, 16), (232, 8), (460, 8)])     # 460: the memory-lean long-clip form
def test_ctx_forward_equals_python_path(Tn, h, monkeypatch):
    """Full DAWN architecture: every kernel family / fallback of the evaluation, bit-identical between the two hosts."""
    from fullsize_cases import KW, build_inputs
    from dawn_pytorch_amd import unet_forward as UF
    from dawn_pytorch_amd.ctx import OPT_LONG_CLIP_FRAMES
    lean = Tn == 460                # the memory-lean form of long clips (default above 4096 frames), switched on at 256 in both hosts
    if lean:
        monkeypatch.setattr(UF, "LONG_CLIP_FRAMES", 256)
    unet = D.DynamicNfUnet3D(default_num_frames=Tn, **KW, init_seed=0).cuda()
    ops, P = unet._ops(), unet.packed()
    fea272, cond, x3 = build_inputs(Tn, h)
    fea272, cond, x3 = fea272[0].cuda().contiguous(), cond[0].cuda().contiguous(), x3[0].cuda().contiguous()
    cs = unet.build_clip(fea272, cond)
    want = unet_forward(ops, P, cs, x3, 500)
    ev = CtxEvaluator(P)
    if lean:
        ev.set_option(OPT_LONG_CLIP_FRAMES, 256)
    clip = ev.prepare_clip(fea272, cond, cs.rcos, cs.rsin)
    got = ev.forward(clip, x3, 500.0)
    assert torch.equal(got, want), float((got - want).abs().max())
    # profiling hook: one entry per conv launch, finite times
    ev.set_option(OPT_PROFILE, 1)
    ev.forward(clip, x3, 500.0)
    ev.set_option(OPT_PROFILE, 0)
    pr = ev.profile_read()
    assert len(pr) > 80 and all(p[3] >= 0 and p[1] > 0 for p in pr)


def test_ctx_sampler_golden_and_bit_identical(tiny):
    g, sd = tiny
    d = load_golden("ddim_tiny.npz")
    unet = tiny_unet(sd)
    S = int(d["S"])
    diff = D.DynamicNfGaussianDiffusion(default_num_frames=12, denoise_fn=unet, num_frames=12, image_size=8,
                                        sampling_timesteps=S, timesteps=1000, loss_type='l2', use_dynamic_thres=True,
                                        null_cond_prob=0.1, ddim_sampling_eta=1.0).cuda()
    fea, bbox, cond = T(d["fea"]).cuda(), T(d["bbox"]).cuda(), T(d["cond"]).cuda()
    noises = [n.cuda() for n in T(d["noises"])]
    want = diff.sample(fea, bbox, cond=cond, cond_scale=1.0, x_init=T(d["x_init"]).cuda(), noises=noises, trace=True)
    qs_py = torch.stack([tr["s"] for tr in diff.last_trace[0]])
    P = unet.packed()
    ev = CtxEvaluator(P)
    fea272 = torch.cat((fea, bbox), 1)[0].contiguous()
    cs = unet.build_clip(fea272, cond[0].contiguous())
    clip = ev.prepare_clip(fea272, cond[0].contiguous(), cs.rcos, cs.rsin)
    steps = ddim_step_scalars({k: getattr(diff, k) for k in ("alphas_cumprod_prev", "sqrt_recip_alphas_cumprod",
                                                              "sqrt_recipm1_alphas_cumprod")}, S, 1.0)
    got, thr = ev.sample(clip, T(d["x_init"]).cuda()[0], steps, noises=[n[0].contiguous() for n in noises], want_thresholds=True)
    assert torch.equal(got, want[0]), float((got - want[0]).abs().max())
    assert torch.equal(thr, qs_py)
    assert float((thr[:, 1].cpu() - T(d["quantiles"]).float()).abs().max()) < 5e-5
    assert log("ctx_tiny_ddim", got[None], T(d["out"])) < 1e-5
    # counter-based noise: same seed, same streams as the Python sampler
    diff.noise_seed = 77
    want2 = diff.sample(fea, bbox, cond=cond, cond_scale=1.0, x_init=T(d["x_init"]).cuda())
    got2 = ev.sample(clip, T(d["x_init"]).cuda()[0], steps, seed=77)
    assert torch.equal(got2, want2[0])
    # the sampler hook of the Python API
    diff.use_ctx = True
    want3 = diff.sample(fea, bbox, cond=cond, cond_scale=1.0, x_init=T(d["x_init"]).cuda())
    assert torch.equal(want3, want2)
