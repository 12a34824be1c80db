"""-m gpu: T-shard THROUGH THE C ABI (include/dawn_hip.h: dawn_shard_comm, dawn_unet_forward_sharded, dawn_sampler_run_sharded) --
what a non-Python host would drive a rank with.  The three exchanges are host callbacks; here they are implemented in-process:

  * world 1 (callbacks with nothing to exchange): bit-identical to the Python-orchestrated sharded path of the same rank;
  * world 2 on ONE GPU: two evaluators in two host threads, the callbacks hand frames / sums to each other through a barrier
    (no RCCL, no gloo) -- the concatenated result equals the unsharded clip (evaluation and a DDIM trajectory with the
    counter-based noise, whose streams are keyed by the global element index)."""
import pytest
import torch

from conftest import load_golden
from test_hip_end2end import tiny_unet, T, log
from inproc_shard import Exchange as _Exchange, run_ranks as _run_ranks
import dawn_pytorch_amd as D
from dawn_pytorch_amd.ctx import CtxEvaluator, ShardCallbacks
from dawn_pytorch_amd.sampler import ddim_step_scalars
from dawn_pytorch_amd.tshard import SimulatedInteriorShard
from dawn_pytorch_amd.unet_forward import unet_forward

pytestmark = pytest.mark.gpu


def _noop_callbacks(rank=0, world=1):
    nop = lambda *a: None                                       # noqa: E731
    return ShardCallbacks(rank, world, nop, nop, nop, nop, nop)


def test_ctx_sharded_world1_bit_identical_to_python_sharded_path(tiny):
    g, sd = tiny
    unet = tiny_unet(sd)
    ops, P = unet._ops(), unet.packed()
    x = T(g["x"]).cuda()
    cond = T(g["cond"]).cuda()
    fea272 = x[0, 3:, 0].contiguous()
    x3, t = x[0, :3].contiguous(), float(g["time"][0])
    Fn = cond.shape[1]
    comm = SimulatedInteriorShard(Fn, world=1, rank=0)          # world 1: no halo, all-reduces are identities
    cs = unet.build_clip(fea272, cond[0].contiguous(), comm=comm, Ttotal=Fn, f0=0)
    want = unet_forward(ops.with_comm(comm), P, cs, x3, int(t))
    ev = CtxEvaluator(P)
    clip = ev.prepare_clip(fea272, cond[0].contiguous(), cs.rcos, cs.rsin)
    got = ev.forward(clip, x3, t, shard=_noop_callbacks())
    assert torch.equal(got, want), float((got - want).abs().max())
    assert log("ctx_sharded_world1_vs_golden", got[None], T(g["y"])) < 2e-5
    # a missing callback is an error code, not a silent no-op
    from dawn_pytorch_amd._lib import DawnHipError
    broken = _noop_callbacks()
    broken.c.allreduce_sum_f64 = type(broken.c.allreduce_sum_f64)()
    with pytest.raises(DawnHipError):
        ev.forward(clip, x3, t, shard=broken)


@pytest.mark.parametrize("world", [2, 3])
def test_ctx_sharded_two_ranks_in_process_equal_unsharded(tiny, world):
    g, sd = tiny
    unet = tiny_unet(sd)                                        # win = 3
    ops, P = unet._ops(), unet.packed()
    Fr = 8                                                      # frames per rank (> win: one neighbour per halo)
    Tt = Fr * world
    gen = torch.Generator().manual_seed(11)
    fea272 = T(g["x"])[0, 3:, 0].contiguous().cuda()
    cond = torch.randn(Tt, T(g["cond"]).shape[2], generator=gen).cuda()
    x3 = torch.randn(3, Tt, 8, 8, generator=gen).cuda()
    unet.update_num_frames(Tt)
    cs_full = unet.build_clip(fea272, cond)
    want = unet_forward(ops, P, cs_full, x3, 500)
    torch.cuda.synchronize()
    ex = _Exchange(world)
    evs = [CtxEvaluator(P) for _ in range(world)]

    def rank_forward(r):
        clip = evs[r].prepare_clip(fea272, cond[r * Fr:(r + 1) * Fr].contiguous())
        return evs[r].forward(clip, x3[:, r * Fr:(r + 1) * Fr].contiguous(), 500.0, shard=ex.callbacks(r))
    got = torch.cat(_run_ranks(world, rank_forward), dim=1)
    err = log(f"ctx_sharded_world{world}_forward_vs_unsharded", got, want)
    assert err < 2e-5 * max(1.0, float(want.abs().max())), err

    # DDIM trajectory: counter-based noise (global element index) + whole-clip quantile through the histogram all-reduces
    S = 3
    diff = D.DynamicNfGaussianDiffusion(default_num_frames=Tt, denoise_fn=unet, num_frames=Tt, image_size=8, sampling_timesteps=S,
                                        timesteps=1000, loss_type='l2', use_dynamic_thres=True, ddim_sampling_eta=1.0).cuda()
    diff.update_num_frames(Tt)
    diff.noise_seed = 77
    want_s = diff.sample(fea272[None, :-4], fea272[None, -4:], cond=cond[None], cond_scale=1.0, x_init=x3[None])[0]
    steps = ddim_step_scalars({k: getattr(diff, k) for k in ("alphas_cumprod_prev", "sqrt_recip_alphas_cumprod",
                                                              "sqrt_recipm1_alphas_cumprod")}, S, 1.0)

    def rank_sample(r):
        clip = evs[r].prepare_clip(fea272, cond[r * Fr:(r + 1) * Fr].contiguous())
        return evs[r].sample(clip, x3[:, r * Fr:(r + 1) * Fr].contiguous(), steps, seed=77, want_thresholds=True,
                             shard=ex.callbacks(r))
    res = _run_ranks(world, rank_sample)
    got_s = torch.cat([o for o, _ in res], dim=1)
    for _, thr in res[1:]:
        assert torch.equal(thr, res[0][1])                     # every rank selected the same whole-clip threshold
    err = log(f"ctx_sharded_world{world}_ddim_vs_unsharded", got_s, want_s)
    assert err < 5e-5, err


def test_ctx_sharded_long_shards_run_the_lean_form(tiny):
    """DAWN_OPT_LONG_CLIP_FRAMES applies to sharded calls too (VERDICT r3 #5 / ADVICE): shards longer than the option run the
    memory-lean form -- qkv of the unfused attention levels per 200-query segment on row windows of the extended buffer, the heads'
    skip recomputed, the heads one after the other -- with a smaller workspace, and still equal the unsharded clip.  Two in-process
    ranks of 230 frames (two segments each, the second one short), window 3."""
    from dawn_pytorch_amd import ctx as C_
    g, sd = tiny
    unet = tiny_unet(sd)
    ops, P = unet._ops(), unet.packed()
    world, Fr = 2, 230
    Tt = Fr * world
    gen = torch.Generator().manual_seed(12)
    fea272 = T(g["x"])[0, 3:, 0].contiguous().cuda()
    cond = torch.randn(Tt, T(g["cond"]).shape[2], generator=gen).cuda()
    x3 = torch.randn(3, Tt, 8, 8, generator=gen).cuda()
    unet.update_num_frames(Tt)
    want = unet_forward(ops, P, unet.build_clip(fea272, cond), x3, 500)
    torch.cuda.synchronize()
    need = {}
    for lean in (False, True):
        ex = _Exchange(world)
        evs = [CtxEvaluator(P) for _ in range(world)]
        if lean:
            for ev in evs:
                ev.set_option(C_.OPT_LONG_CLIP_FRAMES, 100)

        def rank_forward(r):
            clip = evs[r].prepare_clip(fea272, cond[r * Fr:(r + 1) * Fr].contiguous())
            cb = ex.callbacks(r)
            out = evs[r].forward(clip, x3[:, r * Fr:(r + 1) * Fr].contiguous(), 500.0, shard=cb)
            need[(lean, r)] = evs[r]._need[(Fr, 8, 8, evs[r]._policy, (r, world))]
            return out
        got = torch.cat(_run_ranks(world, rank_forward), dim=1)
        err = log(f"ctx_sharded_world2_long_shards_lean{int(lean)}_vs_unsharded", got, want)
        assert err < 2e-5 * max(1.0, float(want.abs().max())), (lean, err)
    assert need[(True, 0)] < need[(False, 0)], need                  # the lean form needs less workspace
